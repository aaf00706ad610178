#!/usr/bin/env bash
# usage: benchv1.sh <label>... : sponza bench (5 steps) of csrc/variants/<label>.so ("main" = the shipped library)
for label in "$@"; do
  lib=""; [ "$label" != main ] && lib=$PWD/raytracing-course-hw-public_amd/csrc/variants/$label.so
  RT_AMD_LIB=$lib python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/bv_${label}_sponza.json 2> gpurun_out/bv_${label}_sponza.err
  python -c "
import json; j=json.load(open('gpurun_out/bv_${label}_sponza.json')); print('$label', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
done

#!/usr/bin/env bash
# usage: benchv.sh <label> [lib path]
label=$1; lib=$2
for wl in sponza s10m; do
  if [ "$wl" = s10m ]; then extra="--workload s10m --steps 2 --warmup 1"; else extra="--steps 5 --warmup 2"; fi
  RT_AMD_LIB=$lib python bench.py --no-cpu-baseline $extra > gpurun_out/bv_${label}_$wl.json 2> gpurun_out/bv_${label}_$wl.err
  python -c "
import json; j=json.load(open('gpurun_out/bv_${label}_$wl.json')); print('$label', '$wl', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
done

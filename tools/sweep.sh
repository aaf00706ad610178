#!/usr/bin/env bash
# usage: sweep.sh variant...   (runs bench at 16 SPP for each variant lib)
for v in "$@"; do
  if [ "$v" = "base" ]; then lib=""; else lib="$PWD/raytracing-course-hw-public_amd/csrc/variants/$v.so"; fi
  r=$(RT_AMD_LIB=$lib timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --spp ${SPP:-16} 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['pipeline']['device_ms_per_step'], j['roofline']['avg_launch_ms'])")
  echo "$v $r"
done

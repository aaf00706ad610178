#!/usr/bin/env bash
# tools_sweep_env.sh VAR val... — bench at 64 SPP for each value of an environment variable (development aid)
var=$1; shift
for v in "$@"; do
  r=$(env $var=$v timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['avg_launch_ms'], j['roofline']['launches_per_step'])")
  echo "$var=$v $r"
done

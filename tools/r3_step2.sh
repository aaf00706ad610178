#!/usr/bin/env bash
# round 3, step 2: the 8-wide production build — tests, then bench of wide against the binary variants
set -e
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_production.py -x -q -s -k wide > $O/s2_tests.log 2>&1 || { tail -60 $O/s2_tests.log; exit 1; }
tail -8 $O/s2_tests.log
for cfg in "--wide" "--wide --bvh device" "--traversal global"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 $cfg > $O/s2_bench_$tag.json 2> $O/s2_bench_$tag.err || { tail -5 $O/s2_bench_$tag.err; exit 1; }
  python - $O/s2_bench_$tag.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=j["roofline"]["pipeline"]
print(sys.argv[1], j["value"], "Msamples/s", "avg_launch_ms", j["roofline"]["avg_launch_ms"], "nodes/cast", p["nodes_per_cast"], "tri/cast", p["tri_tests_per_cast"], "create_s", j["setup_s"])
PY
done

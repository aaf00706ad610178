#!/usr/bin/env bash
# round 3, step 1: production-mode tests + bench of the traversal variants (reference / global best x reference tree / LBVH)
set -e
O=gpurun_out
python -m pytest tests/test_gpu_production.py -x -q -s > $O/s1_tests.log 2>&1 || { tail -40 $O/s1_tests.log; exit 1; }
tail -5 $O/s1_tests.log
for t in reference global; do for b in reference device; do
  python bench.py --no-cpu-baseline --steps 5 --warmup 2 --traversal $t --bvh $b > $O/s1_bench_${t}_${b}.json 2> $O/s1_bench_${t}_${b}.err || { tail -5 $O/s1_bench_${t}_${b}.err; exit 1; }
  python - $O/s1_bench_${t}_${b}.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=j["roofline"]["pipeline"]
print(sys.argv[1], j["value"], "Msamples/s", "avg_launch_ms", j["roofline"]["avg_launch_ms"], "nodes/cast", p["nodes_per_cast"], "tri/cast", p["tri_tests_per_cast"])
PY
done; done

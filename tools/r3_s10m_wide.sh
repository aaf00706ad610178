#!/usr/bin/env bash
O=gpurun_out
for cfg in "" "--wide" "--wide --bvh device" "--traversal global"; do
  tag=$(echo "base$cfg" | tr -d ' -')
  timeout -k 10 500 python bench.py --workload s10m --no-cpu-baseline --steps 2 --warmup 1 $cfg > $O/s10_$tag.json 2> $O/s10_$tag.err || { tail -5 $O/s10_$tag.err; continue; }
  python - $O/s10_$tag.json $tag <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=j["roofline"]["pipeline"]
print(f"\n{sys.argv[2]:22s} {j['value']:8.1f} Msamples/s  extend {j['roofline']['avg_launch_ms']:7.3f} ms  nodes/cast {p['nodes_per_cast']}  tri/cast {p['tri_tests_per_cast']} setup {j['setup_s']}")
PY
done

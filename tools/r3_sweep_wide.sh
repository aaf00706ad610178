#!/usr/bin/env bash
# round 3: parameter sweep of the wide path on S-sponza (bench.py --wide), variants built by tools/variants.sh
O=gpurun_out
V=raytracing-course-hw-public_amd/csrc/variants
run() { # label, lib ('' = shipped), extra env
  label=$1; lib=$2; shift 2
  env $@ ${lib:+RT_AMD_LIB=$lib} python bench.py --no-cpu-baseline --steps 3 --warmup 1 --wide $BENCH_EXTRA > $O/sw_$label.json 2> $O/sw_$label.err || { echo "$label FAILED"; tail -3 $O/sw_$label.err; return; }
  python - $O/sw_$label.json $label <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=j["roofline"]["pipeline"]
print(f"{sys.argv[2]:14s} {j['value']:8.1f} Msamples/s  extend {j['roofline']['avg_launch_ms']:7.3f} ms  nodes/cast {p['nodes_per_cast']}  tri/cast {p['tri_tests_per_cast']}")
PY
}
run base ""
for ct in 0.1 0.2 0.5 1.0; do run ct$ct "" RT_WIDE_COST_TRI=$ct; done
for v in w6 d6 d10 t12 t28 r8 r24 c256; do run $v $V/$v.so; done

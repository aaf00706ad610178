#!/usr/bin/env bash
V=raytracing-course-hw-public_amd/csrc/variants
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for lib in "" $V/s6.so; do for mode in parity wide; do
RT_AMD_LIB=$lib python bench.py --mode $mode --no-extras --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-shipped} $mode', j['value'], 'device ms', j['roofline']['pipeline']['device_ms_per_step'], 'extend', j['roofline']['avg_launch_ms'])"
done; done

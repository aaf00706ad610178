#!/usr/bin/env bash
timeout -k 10 600 python -m pytest tests/test_gpu_production.py -x -q -k "wide" 2>&1 | tail -4
for wl in sponza s10m; do for pk in 0 1; do
RT_WF_PACKET=$pk python bench.py --workload $wl --mode wide --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl packet=$pk', j['value'], 'extend', j['roofline']['avg_launch_ms'], 'device ms', j['roofline']['pipeline']['device_ms_per_step'])"
done; done
python bench.py --workload sponza --mode wide --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sponza default', j['value'])"

#!/usr/bin/env bash
for wl in sponza s10m; do for sm in 0 1 4; do
RT_WF_SORT=$sm python bench.py --workload $wl --mode wide --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl sort=$sm', j['value'], 'extend', j['roofline']['avg_launch_ms'], 'device ms', j['roofline']['pipeline']['device_ms_per_step'])"
done; done

#!/usr/bin/env bash
# PLOC search radius: render speed of the tree it yields (global-best on the binary tree; the wide tree collapsed from it) and build time
for r in ${RADII:-4 8 16}; do for wl in sponza s10m; do for mode in wide global; do
RT_PLOC_RADIUS=$r python bench.py --workload $wl --mode $mode --bvh device --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j['roofline']['pipeline']; print('radius $r $wl $mode', j['value'], 'visits', p['nodes_per_cast'], 'tri', p['tri_tests_per_cast'], 'build s', j['setup_s']['scene_bvh_build'], 'rt_create', j['setup_s']['rt_create_bvh_upload'])"
done; done; done

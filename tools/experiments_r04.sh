#!/usr/bin/env bash
# tools/experiments_r04.sh <name> — the A/B measurements behind profiles/r04_variants.txt (on the GPU box; variants built by tools/variants*.sh)
V=raytracing-course-hw-public_amd/csrc/variants
B="python bench.py --no-extras --no-cpu-baseline --no-config4"
line() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$1', j['value'], 'Msamples/s; closest-hit launch', r['avg_launch_ms'], 'ms; nodes/cast', r['pipeline']['nodes_per_cast'], 'packet', r['packet']['packet_passes'], '/', r['packet']['passes'], r['packet']['lanes_served_per_trip'])"; }
case "$1" in
  occupancy)  # wf_extend_wide at 4 / 5 (shipped) / 6 waves per SIMD; refill threshold 8 / 16 (shipped) / 24 / 32
    for wl in s10m sponza; do for v in "" w4 w6 r8 r24 r32; do lib=${v:+$V/$v.so}; RT_AMD_LIB=$lib $B --workload $wl --mode wide --bvh device --steps 3 --warmup 1 2>/dev/null | line "$wl wide ${v:-shipped}"; done; done ;;
  sortkey)    # ray-order keys on the out-of-cache scene: off / cell+octant+cone / octant+cell+cone (AUTO) / octant+128^3 cell+cone
    for mode in wide parity; do bvh=$([ $mode = wide ] && echo device || echo reference); for sm in 0 4 5 6; do RT_WF_SORT=$sm $B --workload s10m --mode $mode --bvh $bvh --steps 2 --warmup 1 2>/dev/null | line "s10m $mode sort=$sm"; done; done ;;
  packet)     # primary rays as packets on / off, both workloads, both builds (the policy must pick the faster one)
    for wl in sponza s10m; do for mode in parity wide; do bvh=$([ $mode = wide ] && echo device || echo reference); for pk in 0 1 ""; do RT_WF_PACKET=$pk; [ -z "$pk" ] && unset RT_WF_PACKET || export RT_WF_PACKET; $B --workload $wl --mode $mode --bvh $bvh --steps 3 --warmup 1 2>/dev/null | line "$wl $mode packet=${pk:-auto}"; done; unset RT_WF_PACKET; done; done ;;
  *) echo "usage: $0 occupancy|sortkey|packet" ;;
esac

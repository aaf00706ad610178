#!/usr/bin/env bash
# tools/experiments_r03.sh <name> — the measurements behind profiles/r03_*.txt, each runnable on the GPU box (gpurun -- 'bash tools/experiments_r03.sh <name>').
#   modes      parity / global-best / wide x reference tree / device LBVH on S-sponza and S-10M          -> profiles/r03_wide.txt
#   sweep      parameter sweep of the wide kernel (needs tools/variants.sh builds, see the case below)      -> profiles/r03_wide.txt
#   sort       coherence sort on / off in wide mode                                                         -> profiles/r03_wide.txt
#   packet     wide packet kernel for primary rays on / off                                                 -> profiles/r03_wide.txt
#   order      DevNode order in HBM: pre-order / breadth first / sibling pairs (parity mode)                -> profiles/r03_variants.txt
#   shade      wf_shade at 5 and 6 waves per SIMD (needs variant s6: -DRT_SHADE_WAVES_PER_SIMD=6)           -> profiles/r03_variants.txt
#   shard      one GPU's share of a strong-scaled render against the whole render + its kernel timeline     -> profiles/r03_variants.txt
set -e
O=gpurun_out; mkdir -p $O
V=raytracing-course-hw-public_amd/csrc/variants
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=j['roofline']['pipeline']
print('$1', j['value'], 'Msamples/s  closest-hit launch', j['roofline']['avg_launch_ms'], 'ms  visits/cast', p['nodes_per_cast'], ' triangle tests/cast', p['tri_tests_per_cast'], ' rt_create', j['setup_s']['rt_create_bvh_upload'], 's')"; }
B="python bench.py --no-extras --no-cpu-baseline"
case "$1" in
modes)
  for wl in sponza s10m; do for cfg in "--mode parity" "--mode global" "--mode wide" "--mode parity --bvh device" "--mode global --bvh device" "--mode wide --bvh device"; do
    $B --workload $wl $cfg --steps 3 --warmup 1 2>/dev/null | line "$wl $cfg"; done; done ;;
sweep) # tools/variants.sh w6:-DRT_WIDE_WAVES_PER_SIMD=6 d6:-DRT_WIDE_LDS_DEPTH=6 d10:-DRT_WIDE_LDS_DEPTH=10 t12:-DRT_WIDE_TRI_MIN=12 t28:-DRT_WIDE_TRI_MIN=28 r8:-DRT_WIDE_REFILL_MIN=8 r24:-DRT_WIDE_REFILL_MIN=24 c256:-DRT_WIDE_CHUNK=256u
  $B --mode wide --steps 3 --warmup 1 2>/dev/null | line base
  for ct in 0.1 0.2 0.5 1.0; do RT_WIDE_COST_TRI=$ct $B --mode wide --steps 3 --warmup 1 2>/dev/null | line "cost_tri=$ct"; done
  for v in w6 d6 d10 t12 t28 r8 r24 c256; do [ -f $V/$v.so ] && RT_AMD_LIB=$V/$v.so $B --mode wide --steps 3 --warmup 1 2>/dev/null | line $v; done ;;
sort)
  for wl in sponza s10m; do for sm in 0 1 4; do RT_WF_SORT=$sm $B --workload $wl --mode wide --steps 3 --warmup 1 2>/dev/null | line "$wl sort=$sm"; done; done ;;
packet)
  for wl in sponza s10m; do for pk in 0 1; do RT_WF_PACKET=$pk $B --workload $wl --mode wide --steps 3 --warmup 1 2>/dev/null | line "$wl packet=$pk"; done; done ;;
order)
  for wl in s10m sponza; do for om in 0 1 2; do RT_NODE_ORDER=$om $B --workload $wl --mode parity --steps 2 --warmup 1 2>/dev/null | line "$wl node_order=$om"; done; done ;;
shade)
  for lib in "" $V/s6.so; do for mode in parity wide; do RT_AMD_LIB=$lib $B --mode $mode --steps 5 --warmup 2 2>/dev/null | line "lib=${lib:-shipped} $mode"; done; done ;;
shard)
  python tools/shard_timing.py; python tools/shard_timing.py --wide; bash tools/shard_trace.sh ;;
*) sed -n 2,12p "$0" ;;
esac

#!/usr/bin/env bash
# tools/profile_round.sh <workload: sponza|s10m> [tag] [mode: parity|wide|global] [bvh: reference|device] — round deliverables for one bench workload and
# traversal mode, on the GPU box:
#   1. the bench.py run of that workload / mode                         -> gpurun_out/<tag>_bench_<id>.json
#   2. rocprofv3 --kernel-trace --stats of the same command             -> gpurun_out/<tag>_stats_<id>/
#   3. FETCH_SIZE and WRITE_SIZE, one --pmc pass each (counters only)   -> profiles-ready <tag>_hbm_traffic_<id>.json
#   4. SQ / TCP / TCC counter passes at reduced SPP                     -> profiles-ready <tag>_pmc_wf_extend_<id>.json (+ wf_shade)
# <id> = bench.py's workload_id (sponza, sponza-wide, s10m, ...). Every summary is stamped with the hash of the device sources
# (bench.py kernel_source_hash) so a later bench run can tell whether it still describes the kernels it is running. Copy what
# should be judged from gpurun_out/ into profiles/ (tools/install_profiles.sh).
wl=${1:-sponza}; tag=${2:-r04}; mode=${3:-parity}; bvh=${4:-reference}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
id=$wl; [ "$bvh" = device ] && id=$wl-dev; [ "$mode" = wide ] && id=$id-wide; [ "$mode" = global ] && id=$id-gbest
if [ "$wl" = s10m ]; then pmc_spp=8; else pmc_spp=16; fi
extra="--workload $wl --mode $mode --bvh $bvh --no-extras --no-config4 --live-pmc off"
cpu=""; { [ "$wl" = s10m ] || [ "$mode" != parity ]; } && cpu="--no-cpu-baseline"  # the CPU leg is on the default bench line (S-sponza, parity)
python3 $R/bench.py $extra $cpu > $O/${tag}_bench_$id.json 2> $O/${tag}_bench_$id.err; echo "bench exit $?"; tail -c 1500 $O/${tag}_bench_$id.json
cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats_$id -- python3 $R/bench.py $extra --no-cpu-baseline > $O/${tag}_stats_$id.log 2>&1; echo "stats pass exit $?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_pmc_${c}_$id --pmc $c -- python3 $R/bench.py $extra --no-cpu-baseline --steps 1 --warmup 1 > $O/${tag}_pmc_${c}_$id.log 2>&1; echo "$c pass exit $?"
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_pmc_sq${i}_$id --pmc $set -- python3 $R/bench.py $extra --no-cpu-baseline --steps 1 --warmup 0 --spp $pmc_spp > $O/${tag}_pmc_sq${i}_$id.log 2>&1 || echo "sq pass $i failed"
done
cd $R
python3 tools/pmc_summarize.py $wl $tag $pmc_spp $mode $bvh

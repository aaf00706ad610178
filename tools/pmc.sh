#!/usr/bin/env bash
# tools_pmc.sh <tag> <spp> — rocprofv3 PMC passes over bench.py (development aid). Writes gpurun_out/pmc_<tag>.txt
tag=$1; spp=${2:-8}
export TMPDIR=/tmp; R=$PWD; cd /tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp $spp"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$i --pmc $set -- $B > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY > $R/gpurun_out/pmc_$tag.txt
import csv, glob, collections
agg=collections.defaultdict(float); dur={}
for f in glob.glob("$R/gpurun_out/pmc_${tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'render_kernel<0, false>' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
            dur[r['Counter_Name']] = (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
for k in sorted(agg): print(f"{k:36s} {agg[k]:.6g}   (kernel {dur[k]:.1f} ms)")
PY
cat $R/gpurun_out/pmc_$tag.txt

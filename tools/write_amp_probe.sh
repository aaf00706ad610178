#!/usr/bin/env bash
# tools/write_amp_probe.sh <tag> — where do wf_extend's HBM-side writes come from? WRITE_SIZE and the EA write-request split
# with the coherence sort on (hits stored at sorted slots) and off (hits stored at queue position), 16 SPP.
tag=${1:-wa}
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O; cd /tmp
for sort in 1 0; do
  for set in "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum"; do
    name=${tag}_sort${sort}_$(echo $set | cut -d' ' -f1)
    RT_WF_SORT=$sort timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/$name --pmc $set -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --spp 16 > $O/$name.log 2>&1 || echo "$name failed"
  done
done
python3 - <<PY
import csv, glob, collections
for sort in (1, 0):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("$O/${tag}_sort%d_*/*/*counter_collection.csv" % sort):
        for r in csv.DictReader(open(f)):
            if "wf_extend<false>" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("sort", sort, {k: (v, n[k]) for k, v in agg.items()})
PY

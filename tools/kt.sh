#!/usr/bin/env bash
# usage: kt.sh variant  -> per-kernel avg ms via rocprofv3 stats at 64 spp
v=$1; export TMPDIR=/tmp; R=$PWD
if [ "$v" = "base" ]; then lib=""; else lib="$R/raytracing-course-hw-public_amd/csrc/variants/$v.so"; fi
cd /tmp
RT_AMD_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$v -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/kt_$v.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/kt_$v/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if "wf_" in n and "true" not in n: print("$v", n[:60], r["Calls"], float(r["AverageNs"])/1e6)
PY

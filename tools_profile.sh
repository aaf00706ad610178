#!/usr/bin/env bash
# tools_profile.sh <tag> — round deliverables: default bench.py run, the rocprofv3 --kernel-trace --stats summary of the
# same command, and FETCH_SIZE / WRITE_SIZE PMC passes (separate runs) for roofline.traffic. Outputs under gpurun_out/.
tag=$1
export TMPDIR=/tmp; R=$PWD
python3 $R/bench.py > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err; echo "bench exit $?"; tail -c 1500 $R/gpurun_out/bench_$tag.json
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_$tag.log 2>&1; echo "stats pass exit $?"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_$tag --pmc FETCH_SIZE -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $R/gpurun_out/pmcf_$tag.log 2>&1; echo "fetch pass exit $?"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcw_$tag --pmc WRITE_SIZE -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $R/gpurun_out/pmcw_$tag.log 2>&1; echo "write pass exit $?"
python3 - <<PY
import csv, glob
for kind in ("pmcf","pmcw"):
    for f in glob.glob("$R/gpurun_out/%s_$tag/*/*counter_collection.csv" % kind):
        for r in csv.DictReader(open(f)):
            if 'render_kernel<0, false>' in r['Kernel_Name']:
                print(kind, r['Counter_Name'], r['Counter_Value'], "dispatch", r['Dispatch_Id'])
for f in glob.glob("$R/gpurun_out/prof_$tag/*/*kernel_stats.csv"):
    print(open(f).read()[:900])
PY

#!/usr/bin/env bash
# tools_profile.sh <tag> — round deliverables: the default bench.py run, the rocprofv3 --kernel-trace --stats summary of
# the same command, and FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, counters only) for roofline.traffic of the
# dominant kernel. Outputs under gpurun_out/; copy what should be judged into profiles/.
tag=$1
export TMPDIR=/tmp; R=$PWD
python3 $R/bench.py > $R/gpurun_out/bench_$tag.json 2> $R/gpurun_out/bench_$tag.err; echo "bench exit $?"; tail -c 2200 $R/gpurun_out/bench_$tag.json
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_$tag.log 2>&1; echo "stats pass exit $?"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_$tag --pmc FETCH_SIZE -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $R/gpurun_out/pmcf_$tag.log 2>&1; echo "fetch pass exit $?"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcw_$tag --pmc WRITE_SIZE -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > $R/gpurun_out/pmcw_$tag.log 2>&1; echo "write pass exit $?"
python3 - <<PY
import csv, glob, json
kern = "wf_extend<false>"
tot = {}
n = {}
for kind in ("pmcf", "pmcw"):
    for f in glob.glob("$R/gpurun_out/%s_$tag/*/*counter_collection.csv" % kind):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
print(tot, n)
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    fetch = tot["FETCH_SIZE"] / n["FETCH_SIZE"]
    write = tot["WRITE_SIZE"] / n["WRITE_SIZE"]
    out = {"workload": "S-sponza 1000x1000x64 n=262144", "kernel": "wf_extend", "launches": n["FETCH_SIZE"],
           "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
           "correction": "gfx950: FETCH_SIZE reports half of the fetched bytes -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is; x1024 (KB)",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over python3 bench.py --no-cpu-baseline --steps 1 --warmup 0, averaged over the wf_extend<false> dispatches"}
    json.dump(out, open("$R/gpurun_out/hbm_traffic_$tag.json", "w"), indent=1)
    print(out)
for f in glob.glob("$R/gpurun_out/prof_$tag/*/*kernel_stats.csv"):
    print(open(f).read()[:1500])
PY

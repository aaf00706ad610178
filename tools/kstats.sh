#!/usr/bin/env bash
# tools/kstats.sh <label> [bench args...] — rocprofv3 kernel-trace stats of one short bench run; prints the wf_* kernels' average durations
label=$1; shift
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$label -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $O/ks_$label.log 2>&1
f=$(ls $O/ks_$label/*/*kernel_stats.csv | head -1)
python3 - "$f" "$label" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "wf_" in n or "radix" in n.lower():
        short = n.split("(")[0].split("::")[-1][:40] if "wf_" in n else "rocprim " + ("onesweep_iter" if "onesweep_iteration" in n else "other")
        print(sys.argv[2], f"{short:42s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e6:8.3f} ms  total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
PY

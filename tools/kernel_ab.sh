#!/usr/bin/env bash
# tools/kernel_ab.sh <out-tag> <mode: parity|wide> <workload> [variant.so] — development aid: per-kernel times (rocprofv3 --kernel-trace --stats)
# of one bench step, for the shipped library or a tools/variants.sh variant. Prints kernel, calls, average microseconds.
set -e
tag=$1; mode=$2; wl=$3; lib=$4
R="$(cd "$(dirname "$0")/.." && pwd)"; O=$R/gpurun_out
[ -n "$lib" ] && export RT_AMD_LIB=$R/raytracing-course-hw-public_amd/csrc/variants/$lib.so
extra="--workload $wl --mode $mode --no-extras --no-cpu-baseline --no-config4 --live-pmc off --steps 2 --warmup 1"
[ "$mode" = wide ] && extra="$extra --bvh device"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kab_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kab_$tag -- python3 $R/bench.py $extra > $O/kab_$tag.log 2>&1
python3 - $O/kab_$tag "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:9]:
    n = r["Name"]; n = n[n.find("wf_"):][:40] if "wf_" in n else n[:40]
    print(f"{sys.argv[2]:18s} {n:42s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:10.1f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
PY

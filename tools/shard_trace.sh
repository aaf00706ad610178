#!/usr/bin/env bash
# kernel timeline of the 1/8-shard render (tools/shard_timing.py): where do the per-bounce fixed costs go?
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/shardtrace -- python3 $R/tools/shard_timing.py $1 > $O/shardtrace.log 2>&1
cd $R; f=$(ls $O/shardtrace/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last render of the run = the last wf_generate .. film kernel sequence (shard 1/8)
gen = [i for i, r in enumerate(rows) if "wf_generate" in r["Kernel_Name"]]
a = gen[-1]
seq = rows[a:]
end = max(i for i, r in enumerate(seq) if "film_kernel" in r["Kernel_Name"] or "wf_resolve" in r["Kernel_Name"])
seq = seq[: end + 1]
t0, t1 = int(seq[0]["Start_Timestamp"]), int(seq[-1]["End_Timestamp"])
busy = collections.defaultdict(float); cnt = collections.Counter()
gap = 0.0; prev_end = t0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    short = "rocprim" if "rocprim" in n else n.split("(")[0].split("::")[-1].split("<")[0]
    busy[short] += (e - s) / 1e6; cnt[short] += 1
    if s > prev_end: gap += (s - prev_end) / 1e6
    prev_end = max(prev_end, e)
print(f"last render: span {(t1 - t0) / 1e6:.2f} ms, idle gaps {gap:.2f} ms, kernels {len(seq)}")
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {cnt[k]:3d} launches {v:8.3f} ms")
PY

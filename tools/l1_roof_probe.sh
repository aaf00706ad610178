#!/usr/bin/env bash
# tools/l1_roof_probe.sh — the vector-L1 (TCP) access-rate roof for wf_extend's access pattern: tools/ubench/gather64 on tables that
# stay in L2 / Infinity Cache (8-64 MiB, like S-sponza's 8.5 MB of nodes), timed, then once more under rocprofv3 for
# TCP_TOTAL_CACHE_ACCESSES and TCP_TCC_READ_REQ per launch (accesses per record, accesses per clock per CU).
cd "$(dirname "$0")/ubench" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 gather64.hip -o gather64 || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
./gather64 8 16 64 256 | tee $O/l1_roof_gather.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/l1_roof_pmc --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -- $GRAFT_REPO_ROOT/tools/ubench/gather64 8 > $O/l1_roof_pmc.log 2>&1
python3 - $O/l1_roof_pmc <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:24]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
for k, c in acc.items():
    n = max(1, cnt[k]); recs = 2048 * 256 * 256
    print(k, "launches", n, "TCP accesses/record %.2f" % (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / n / recs), "TCP->TCC reads/record %.2f" % (c["TCP_TCC_READ_REQ_sum"] / n / recs),
          "accesses per clock per CU %.3f" % (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (c["GRBM_GUI_ACTIVE"] / 8.0) / 256.0))
PY

#!/usr/bin/env python3
"""tools/pmc_summarize.py <workload> <tag> <pmc_spp> [mode] — turn the rocprofv3 outputs of tools/profile_round.sh (under
gpurun_out/) into the small stamped summaries bench.py reads from profiles/:
    <tag>_hbm_traffic_<wl>.json     FETCH_SIZE / WRITE_SIZE per wf_extend launch, corrected as MI355X_MICROARCH.md says
    <tag>_pmc_<kernel>_<wl>.json    SQ / TCP / TCC counters of wf_extend and wf_shade + derived ratios + a one-line limiter
    <tag>_kernel_stats_<wl>.csv     the --stats table of the default bench command
and prints them. Raw counter sums are kept in the JSON next to every derived figure."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash, WORKLOADS)

wl_name, tag, pmc_spp = sys.argv[1], sys.argv[2], int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "parity"
bvh = sys.argv[5] if len(sys.argv) > 5 else "reference"
wl = wl_name + ("-dev" if bvh == "device" else "") + bench.MODES[mode]["suffix"]  # = bench.py's workload_id: names the output files
O = os.path.join(ROOT, "gpurun_out")
W = bench.WORKLOADS[wl_name]
# the closest-hit kernels of this mode (every one of their launches is what bench.py times as "the dominant kernel")
EXTEND = ("wf_extend_wide<false", "wf_extend_wide_packet<false") if mode == "wide" else ("wf_extend<false", "wf_extend_packet<false")
sha = bench.kernel_source_hash()
workload = f"{W['label']} {W['width']}x{W['height']} n={W['triangles']} mode={mode} bvh={bvh}"


def collect(dirglob, kernel):
    agg = collections.defaultdict(float)
    dur = collections.defaultdict(float)
    seen = collections.defaultdict(set)
    for f in glob.glob(os.path.join(O, dirglob, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in ((kernel,) if isinstance(kernel, str) else kernel)):
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen[r["Counter_Name"]]:
                    seen[r["Counter_Name"]].add(r["Dispatch_Id"])
                    dur[r["Counter_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return agg, dur, {k: len(v) for k, v in seen.items()}


# ---- HBM traffic of wf_extend (full-SPP passes of the default command: same launch structure as the bench)
# both closest-hit kernels: wf_extend<false> (bounces >= 1, and bounce 0 when packets do not pay) and wf_extend_packet<false> (primary
# rays), i.e. the launches bench.py times as "the dominant kernel" (one HIP-event pair per closest-hit launch)
agg, dur, n = collect(f"{tag}_pmc_*_SIZE_{wl}", EXTEND)
if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
    fetch = agg["FETCH_SIZE"] / n["FETCH_SIZE"]
    write = agg["WRITE_SIZE"] / n["WRITE_SIZE"]
    out = {
        "workload": f"{workload} spp={W['spp_per_gpu']}", "kernel": " + ".join(EXTEND) + "...> (every closest-hit launch)", "kernel_src_sha16": sha, "launches": n["FETCH_SIZE"],
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 64 B per L2 read request although a request moves a 128-B line ->
        # doubled, as the guide prescribes. The guide also says "other access widths are uncalibrated: calibrate on a known
        # byte count in your own access pattern": tools/ubench/gather64.hip (profiles/r02_gather64_calibration.txt) shows
        # ONE request per random 64-B record as well as per 128-B record, and a ceiling of ~55 G requests/s for either, so
        # for wf_extend's <= 64-B gathers the REQUEST rate against that measured ceiling is the meaningful utilisation;
        # both are reported.
        "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "hbm_bytes_per_launch_fetch_x1": (fetch + write) * 1024,
        "read_requests_per_launch": fetch * 1024 / 64.0, "request_roof_Greq_s": 55.0,
        "request_roof_source": "profiles/r02_gather64_calibration.txt: random 64-B or 128-B records from 0.5-2 GiB tables complete at 52-57 G requests/s",
        "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
        "avg_launch_ms_under_pmc": dur["FETCH_SIZE"] / n["FETCH_SIZE"],
        "correction": "gfx950: FETCH_SIZE reports 64 B per 128-B read request -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is; x1024 (KB)",
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over python3 bench.py (one warm-up render first: the packet policy has settled) --workload {wl_name} --mode {mode} --bvh {bvh} --no-extras --no-cpu-baseline --steps 1 --warmup 1, averaged over the closest-hit dispatches",
    }
    json.dump(out, open(os.path.join(O, f"{tag}_hbm_traffic_{wl}.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
else:
    print("no FETCH_SIZE/WRITE_SIZE data", dict(agg))

# ---- SQ / TCP / TCC counters
for kern, short in ((("wf_extend_wide<false",) if mode == "wide" else ("wf_extend<false",), "wf_extend"), (("wf_extend_wide_packet<false",) if mode == "wide" else ("wf_extend_packet<false",), "wf_extend_packet"), (("wf_shade<false",), "wf_shade")):
    agg, dur, n = collect(f"{tag}_pmc_sq*_{wl}", kern)
    if not agg:
        print("no SQ data for", kern)
        continue
    g = lambda k: agg.get(k, 0.0)  # noqa: E731
    simds = 256 * 4
    d = {"workload": workload, "spp": pmc_spp, "kernel": kern[0] + "...>", "kernel_src_sha16": sha, "dispatches": max(n.values()),
         "kernel_ms_sum_under_pmc": max(dur.values()), "raw": {k: agg[k] for k in sorted(agg)}}
    # SQ_BUSY_CYCLES is summed over the SQs (one per shader engine: 32); SQ_WAVE_CYCLES etc. are in quad-cycles (x4)
    if g("SQ_BUSY_CYCLES") and g("SQ_INSTS_VALU"):
        cycles_per_simd = g("SQ_BUSY_CYCLES") / 32.0  # busy cycles of an average SQ = kernel-resident cycles
        d["valu_insts_per_simd"] = g("SQ_INSTS_VALU") / simds
        d["busy_cycles"] = cycles_per_simd
        d["valu_busy"] = round(2.0 * g("SQ_INSTS_VALU") / simds / cycles_per_simd, 4)  # a wave64 VALU instruction issues over 2 cycles
        d["lanes_per_valu"] = round(g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU"), 2) if g("SQ_THREAD_CYCLES_VALU") else None
    if g("SQ_WAVE_CYCLES"):
        d["wait_any_frac"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4)
        d["wait_inst_any_frac"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4)
    if g("SQ_INSTS_VALU") and g("SQ_INSTS_SALU"):
        d["salu_per_valu"] = round(g("SQ_INSTS_SALU") / g("SQ_INSTS_VALU"), 3)
    if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
        d["l1_hit"] = round(1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum"), 4)
        # vector-L1 (TCP) access rate against its measured roof (profiles/r02_l1_roof.txt: a CU's L1 retires <= ~0.98 tag
        # accesses per clock; a 16-byte-per-lane load costs one access per lane whose line differs from its neighbours').
        # Shader clock from the pass that carries GRBM_GUI_ACTIVE (summed over the 8 XCDs) and its own kernel time.
        if g("GRBM_GUI_ACTIVE") and dur.get("GRBM_GUI_ACTIVE") and dur.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            clk_per_ms = g("GRBM_GUI_ACTIVE") / 8.0 / dur["GRBM_GUI_ACTIVE"]
            d["shader_clock_ghz"] = round(clk_per_ms / 1e6, 3)
            d["l1_accesses_per_clk_per_cu"] = round(g("TCP_TOTAL_CACHE_ACCESSES_sum") / 256.0 / (dur["TCP_TOTAL_CACHE_ACCESSES_sum"] * clk_per_ms), 4)
            d["l1_roof_accesses_per_clk_per_cu"] = 0.98
            d["l1_frac"] = round(d["l1_accesses_per_clk_per_cu"] / 0.98, 4)
            d["l1_roof_source"] = "profiles/r02_l1_roof.txt (tools/l1_roof_probe.sh: gather of 128-B records, 16-B loads, L2-resident table)"
            d["l1_miss_rate_Greq_s"] = round(g("TCP_TCC_READ_REQ_sum") / dur["TCP_TCC_READ_REQ_sum"] / 1e6, 2) if dur.get("TCP_TCC_READ_REQ_sum") else None
            if g("TCP_PENDING_STALL_CYCLES_sum"):
                d["l1_pending_stall_frac"] = round(g("TCP_PENDING_STALL_CYCLES_sum") / 256.0 / (dur["TCP_PENDING_STALL_CYCLES_sum"] * clk_per_ms), 4)
    if g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        d["l2_hit"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
    if g("TCC_EA0_WRREQ_sum"):
        d["ea_write_bytes"] = (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum")) * 32 + g("TCC_EA0_WRREQ_64B_sum") * 64
        d["ea_write_64B_share"] = round(g("TCC_EA0_WRREQ_64B_sum") / g("TCC_EA0_WRREQ_sum"), 4)
    vb, la, wa = d.get("valu_busy"), d.get("lanes_per_valu"), d.get("wait_any_frac")
    if vb is not None:
        d["limiter"] = (f"latency/issue: VALU issue {vb * 100:.0f} % of slots at {la} of 64 lanes, waves waiting {wa * 100 if wa else 0:.0f} % of their cycles, "
                        f"L2 hit {d.get('l2_hit')}; not HBM bandwidth (see hbm_frac)")
        if d.get("l1_frac") is not None and d["l1_frac"] >= 0.6:
            d["limiter"] = (f"vector-L1 access rate: {d['l1_accesses_per_clk_per_cu']} tag accesses per clock per CU = {d['l1_frac']:.2f} of the measured 0.98 roof "
                            f"({'4 accesses per packed 64-B wide node' if mode == 'wide' else '4 accesses per 64-B node'} per lane), {d.get('l1_pending_stall_frac', 0) * 100:.0f} % of L1 cycles stalled on pending misses; "
                            f"VALU issue {vb * 100:.0f} % at {la} of 64 lanes, waves waiting {wa * 100 if wa else 0:.0f} %, L1 hit {d.get('l1_hit')}, L2 hit {d.get('l2_hit')}; "
                            "not HBM bandwidth (see hbm_frac)")
    path = os.path.join(O, f"{tag}_pmc_{short}_{wl}.json")
    json.dump(d, open(path, "w"), indent=1)
    print(short, json.dumps({k: v for k, v in d.items() if k != "raw"}, indent=1))

for f in glob.glob(os.path.join(O, f"{tag}_stats_{wl}", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(O, f"{tag}_kernel_stats_{wl}.csv"))
    rows = list(csv.reader(open(f)))
    for r in rows[:7]:
        print([c[:60] for c in r])

#!/usr/bin/env python3
"""bench.py — headline benchmark of the render-loop hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload sponza|s10m] [--scaling auto|weak|strong] [--mode parity|global|wide]

Metric (BASELINE.json): Msamples/s, whole job.

Workloads (the reference ships no Sponza asset, so both are the deterministic synthetic sets it names):
  sponza (default, BASELINE config 3/4): "S-sponza" = 262 144 random triangles in a closed 40x16x20 room + 12 wall
         triangles + 16 emissive ceiling triangles, 16 procedural 1024^2 RGBA8 texture sets, 66 materials, white
         environment, ray_depth 8; 1000x1000, 64 SPP per GPU (config 3), or 1000 SPP in all (config 4, --scaling strong).
  s10m   (BASELINE config 5): "S-10M" = the same recipe with 10^7 triangles (vertex offsets +-0.03), 2048x2048,
         32 SPP per GPU (256 SPP on 8 GPUs). Working set (nodes 0.64 GB + triangles 0.48 GB + attributes 1.28 GB) is far
         beyond the 256 MiB Infinity Cache: the configuration where HBM traffic / time / 8 TB/s is a real fraction.
With no --workload the line describes S-sponza (config 3) and, on one GPU, carries config 5 as well: `extra_workloads.s10m`
is the same measurement on S-10M taken in the same invocation (its CPU leg is skipped: the oracle's 10^7-triangle BVH build
alone takes longer than the whole bench should).

One step = one pass of the hot path over one batch = one full render of the image: rt_render_rgb8() through the C-ABI
(render + the film on the device, i.e. what run_raytracer leaves in the reference's Image) with the image resident in
HBM (RT_FLAG_DEVICE_FB) + for N > 1 the RCCL gather of the rgb8 image to rank 0. Scene upload and BVH build happen once
before the timed region, like the reference's RaytracerStaticContext (raytracer.h:633) precedes its pixel loop.

Traversal modes (--mode; the headline is ALWAYS parity unless --mode says otherwise, the others are reported beside it under
"production" on one GPU):
  parity  the reference's tree and traversal order (bvh.h:195-235, near-local pruning): hits, event counters and image are the
          oracle's bit for bit. The headline.
  global  same tree, every box culled against the global best hit (RT_FLAG_GLOBAL_BEST).
  wide    the 8-wide BVH with quantised child boxes (RT_BUILD_WIDE): the production build.

N > 1, two ways to launch, same sharding (interleaved 8-row tiles, block b -> GPU b % N, scene replicated per GPU):
  * python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...   one rank per GPU, dist.gather over RCCL;
  * python bench.py --gpus N                                                   (WORLD_SIZE unset) ONE process: the library's own
    multi-GPU scene (rt_create_on: a replica per GPU + ncclCommInitAll, csrc/rt_group.cpp) renders and gathers.
Either way the line says n_gpus = N only if N GPUs really took part; fewer visible devices or a rank-count mismatch exit non-zero.
--scaling weak (default, = auto): SPP = 64 x N (per-GPU work fixed), at every N: the driver's 1 -> 8 curve is one regime. --scaling strong:
BASELINE config 4 exactly, 1000x1000 at 1000 SPP in all, as the headline. Whatever the headline, a sponza run also carries `config4`: one
timed step of config 4 (1000 SPP in all over the N GPUs taking part) measured after the headline, so a strong-scaling curve exists beside it.

The "roofline" object (dominant kernel: the closest-hit kernel, wf_extend / wf_extend_packet / wf_extend_wide):
  achieved / frac / traffic   HBM-SIDE bytes per launch (rocprofv3 PMC passes kept under profiles/, FETCH_SIZE doubled + WRITE_SIZE
                       as MI355X_MICROARCH.md prescribes for gfx950, stamped with the hash of the device sources they were measured
                       on and used only while that hash matches this run's sources) / the live HIP-event launch duration,
                       against the nominal 8 TB/s. frac is therefore a real fraction of the memory roofline (< 1); null with
                       the reason in traffic_source when no matching profile exists. frac_fetch_x1 is the same with FETCH_SIZE
                       counted once: tools/ubench/gather64.hip shows this kernel's <= 64-byte gathers cost ONE request each,
                       so x1 is what the workload's own calibration supports and x2 is the guide's upper bound.
  algorithmic_GBps / algorithmic_over_peak   SURVEY 8d's ALGORITHMIC bytes (reference-layout record sizes x event counts from the
                       instrumented kernel variant; cache hits are NOT subtracted) / the same launch time. Exceeds 1 when the
                       working set is cache resident: a statement about work, not about HBM.
  binding_roof         the roof that actually binds the kernel in the committed counter profile (vector-L1 access rate, or the
                       HBM request rate for random gathers), its fraction and source.
  request_frac, l1_frac, limiter, pmc   as measured in the same committed profile.
"cpu_baseline": the CPU oracle (port of the reference algorithm, byte-identical to the reference binary on the
fixtures; profiles/r03_cpu_baseline_crosscheck.json times both) on this box's host cores on a bounded sample; rank 0, N = 1 only.
"""
import argparse
import hashlib
import importlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHARD_ROWS = 8
SEED = 0x5EED5EED
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
PROFILE_ROUND = "r04"
CONFIG4_SPP = 1000

WORKLOADS = {
    "sponza": dict(label="S-sponza", width=1000, height=1000, spp_per_gpu=64, triangles=262144, offset=0.15, tex_size=1024, cpu_share=4,
                   metric="Msamples/sec (whole node) on Sponza 1000x1000"),
    "s10m": dict(label="S-10M", width=2048, height=2048, spp_per_gpu=32, triangles=10_000_000, offset=0.03, tex_size=1024, cpu_share=256,
                 metric="Msamples/sec (whole node) on synthetic 10M-triangle scene 2048x2048"),
}
# everything that decides what the closest-hit kernels do and how they are launched (kernels, layouts, the tree builders, the
# launch geometry and workspace policy in rt_scene.cpp, the compiler flags): ONE list, csrc/device_sources.txt, hashed here and by
# the Makefile, which builds the hash into librt_amd.so (rt_source_stamp)
CSRC = os.path.join(ROOT, "raytracing-course-hw-public_amd", "csrc")
DEVICE_SOURCES = tuple(ln.strip() for ln in open(os.path.join(CSRC, "device_sources.txt")) if ln.strip() and not ln.startswith("#"))
MODES = {
    "parity": dict(suffix="", wide=False, gbest=False, text="reference tree, reference order and pruning (parity mode)"),
    "global": dict(suffix="-gbest", wide=False, gbest=True, text="reference tree, global-best pruning (production)"),
    "wide": dict(suffix="-wide", wide=True, gbest=False, text="8-wide quantised BVH, global-best culling, octant order (production build RT_BUILD_WIDE)"),
}


def tuning_from_env(env=None):
    """(render tuning kwargs, build option kwargs) from the development variables the sweep scripts under tools/ set. Since ABI 4 the
    library reads NO environment variable: this file (and the CLI, csrc/host/main.cpp) turns them into rt_params / rt_build_options fields."""
    env = os.environ if env is None else env
    render, build = {}, {}
    if "RT_WF_SORT" in env:
        render["sort_mode"] = int(env["RT_WF_SORT"]) + 1  # RT_SORT_OFF = 1, then the keys in RT_SORT_* order
    if "RT_WF_PACKET" in env:
        render["packet_mode"] = 2 if int(env["RT_WF_PACKET"]) else 1  # RT_PACKET_ON / RT_PACKET_OFF
    if "RT_WF_PACKET_MIN" in env:
        render["packet_min_lanes"] = float(env["RT_WF_PACKET_MIN"])
    if "RT_WF_MAX_PATHS" in env:
        render["max_paths"] = int(float(env["RT_WF_MAX_PATHS"]))
    if env.get("RT_DEVICE_BUILDER") == "lbvh":
        build["device_builder"] = 1
    for var, field, conv in (("RT_PLOC_RADIUS", "ploc_radius", int), ("RT_LBVH_LEAF", "lbvh_leaf_tris", int), ("RT_NODE_ORDER", "node_order", int),
                             ("RT_WIDE_ORDER", "wide_order", int), ("RT_WIDE_COST_NODE", "wide_cost_node", float), ("RT_WIDE_COST_TRI", "wide_cost_tri", float)):
        if var in env:
            build[field] = conv(env[var])
    return render, build


def kernel_source_hash() -> str:
    """sha256 over the device-side sources: stamps PMC profiles so that a stale one is never quoted."""
    h = hashlib.sha256()
    for rel in DEVICE_SOURCES:
        with open(os.path.join(CSRC, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(st: dict, n_pixels: int) -> float:
    """layout-independent event counts x the reference's record sizes. Cache hits do not reduce it."""
    trav = st["box_tests"] * 24 + st["nodes_visited"] * 16 + st["tri_tests"] * 36
    light = st["light_box_tests"] * 24 + st["light_nodes"] * 16 + st["light_tri_tests"] * 36
    shade = st["shaded_hits"] * (96 + 72) + st["texel_fetches"] * 16
    return float(trav + light + shade + 12 * n_pixels)


def effective_cores() -> int:
    """Host threads this process may really use: CPU affinity capped by the cgroup CPU quota (the GPU box exposes 256
    logical CPUs but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def load_profile(name: str, workload_id: str, src_hash: str):
    """A committed PMC summary (profiles/<round>_<name>_<workload>.json) if it describes THIS workload and THESE device
    sources; (None, reason) otherwise."""
    rel = os.path.join("profiles", f"{PROFILE_ROUND}_{name}_{workload_id}.json")
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None, f"{rel}: absent"
    try:
        j = json.load(open(path))
    except Exception as e:  # a damaged file is reported, never guessed around
        return None, f"{rel}: unreadable ({e})"
    if j.get("kernel_src_sha16") != src_hash:
        return None, f"{rel}: stale (measured on device sources {j.get('kernel_src_sha16')}, this run is {src_hash})"
    j["_file"] = rel
    return j, rel


def closest_hit_kernels(mode: str):
    """Name fragments (as rocprofv3 prints them) of the closest-hit kernels of a traversal mode: the launches bench.py times as the dominant kernel."""
    return ("wf_extend_wide<false", "wf_extend_wide_packet<false") if MODES[mode]["wide"] else ("wf_extend<false", "wf_extend_packet<false")


def live_pmc(child_args, mode: str, timeout_s: float = 170.0):
    """Counter evidence for THIS run (the committed PMC summaries describe an earlier run). Three child processes of this very
    command (headline workload only, one warm-up + one timed render) under `rocprofv3 --kernel-trace --pmc <set>`, one counter set per pass as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE, WRITE_SIZE, and the SQ set behind valu_busy / lanes_per_valu. Returns (dict, source text) or
    (None, reason). Children are plain child processes started from a process that has initialised the GPU (fork + exec of rocprofv3, which then
    starts python3 itself): nothing is exec'ed in place."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found on this box"
    kernels = closest_hit_kernels(mode)
    sets = {"FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"], "SQ": ["SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]}
    agg, n_disp, ms = {}, {}, {}
    t_start = time.time()
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for tag, counters in sets.items():
            left = timeout_s - (time.time() - t_start)
            if left < 20:
                return None, f"live PMC passes ran out of their {timeout_s:.0f} s budget before {tag}"
            out_dir = os.path.join(td, tag)
            cmd = [prof, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "--pmc", *counters, "--", sys.executable, os.path.join(ROOT, "bench.py"), *child_args]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=left)
            except subprocess.TimeoutExpired:
                return None, f"live PMC pass {tag} timed out"
            if r.returncode != 0:
                return None, f"live PMC pass {tag} failed (exit {r.returncode}): {(r.stderr or r.stdout)[-200:]!r}"
            seen = set()
            for f in glob.glob(os.path.join(out_dir, "*", "*counter_collection.csv")):
                for row in csv.DictReader(open(f)):
                    if any(k in row["Kernel_Name"] for k in kernels):
                        agg[row["Counter_Name"]] = agg.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                        key = (tag, row["Dispatch_Id"])
                        if key not in seen:
                            seen.add(key)
                            n_disp[tag] = n_disp.get(tag, 0) + 1
                            ms[tag] = ms.get(tag, 0.0) + (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
            if not n_disp.get(tag):
                return None, f"live PMC pass {tag}: no closest-hit dispatch in the counter file"
    fetch = agg["FETCH_SIZE"] / n_disp["FETCH_SIZE"] * 1024.0  # KB per launch -> bytes
    write = agg["WRITE_SIZE"] / n_disp["WRITE_SIZE"] * 1024.0
    out = {"hbm_bytes_per_launch": 2 * fetch + write, "hbm_bytes_per_launch_fetch_x1": fetch + write, "read_requests_per_launch": fetch / 64.0,
           "launches_counted": n_disp["FETCH_SIZE"], "avg_launch_ms_under_pmc": ms["FETCH_SIZE"] / n_disp["FETCH_SIZE"], "seconds": round(time.time() - t_start, 1)}
    if agg.get("SQ_BUSY_CYCLES") and agg.get("SQ_INSTS_VALU"):
        cycles_per_simd = agg["SQ_BUSY_CYCLES"] / 32.0  # one SQ per shader engine; a wave64 VALU instruction issues over 2 cycles (tools/pmc_summarize.py)
        out["valu_busy"] = round(2.0 * agg["SQ_INSTS_VALU"] / (256 * 4) / cycles_per_simd, 4)
        out["lanes_per_valu"] = round(agg["SQ_THREAD_CYCLES_VALU"] / agg["SQ_INSTS_VALU"], 2) if agg.get("SQ_THREAD_CYCLES_VALU") else None
        out["wait_any_frac"] = round(agg["SQ_WAIT_ANY"] / agg["SQ_WAVE_CYCLES"], 4) if agg.get("SQ_WAVE_CYCLES") else None
    return out, (f"measured in THIS run: child processes of the same command under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ set (separate passes, "
                 f"{out['launches_counted']} closest-hit launches counted, {out['seconds']} s); FETCH_SIZE x2 + WRITE_SIZE per MI355X_MICROARCH.md")


def measured_stream_peak():
    """Best streaming-read rate of the box class from the committed microbenchmark record (nominal peak stays 8 TB/s)."""
    for rnd in (PROFILE_ROUND, "r02", "r01"):
        rel = os.path.join("profiles", f"{rnd}_hbm_stream.txt")
        try:
            vals = [float(v) for v in re.findall(r"read\s+([0-9.]+)\s*GB/s", open(os.path.join(ROOT, rel)).read())]
            if vals:
                return max(vals), rel
        except OSError:
            continue
    return None, None


def resolve_launch(gpus: int, env=None, visible_devices=None):
    """How `--gpus N` is carried out. Returns (launcher, rank, local_rank, world): launcher is "single", "torchrun" (one process
    per GPU, started by torch.distributed.run) or "group" (ONE process, the library's own multi-GPU scene). Raises SystemExit
    when N GPUs cannot take part: a line claiming n_gpus = N must never come from fewer devices."""
    env = os.environ if env is None else env
    world = int(env.get("WORLD_SIZE", "1"))
    rank, local_rank = int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0"))
    if gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if world > 1:
        if world != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {gpus}")
        return "torchrun", rank, local_rank, world
    if gpus == 1:
        return "single", 0, local_rank, 1
    if visible_devices is not None and visible_devices < gpus:
        raise SystemExit(f"--gpus {gpus} but only {visible_devices} GPU(s) visible: refusing to report a {gpus}-GPU number from fewer devices")
    return "group", 0, 0, 1


def resolve_spp(scaling: str, gpus: int, wl: dict, explicit_spp: int):
    """(samples per pixel of the whole job, "weak" | "strong")."""
    if explicit_spp > 0:
        return explicit_spp, ("strong" if scaling == "strong" else "weak")
    if scaling == "auto":
        scaling = "weak"  # ONE regime for the whole 1 -> 8 curve the driver assembles; config 4 (strong, 1000 SPP) rides along as `config4` at every N
    if scaling == "strong":
        return CONFIG4_SPP, "strong"  # BASELINE config 4: 1000 SPP in all, whatever N
    return wl["spp_per_gpu"] * gpus, "weak"


def make_scene(rt, wl, n_tri, tex_size, aspect):
    return rt.scenegen.room_scene(n_tri, seed=SEED, tex_size=tex_size, n_tex_sets=16, n_materials=64, n_lights=16,
                                  light_strength=20.0, alpha_fraction=0.02, offset=wl["offset"],
                                  camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9, aspect=aspect))


class Runner:
    """One device scene (one traversal mode) of one workload: timed steps + the roofline record of its closest-hit kernel."""

    def __init__(self, rt, torch, dist, scene, wl_name, mode, W, H, spp, launcher, rank, world, n_gpus, local_rank, backend, film, full_size, device_bvh=False, all_gather=False):
        """`full_size`: True (the workload's own geometry at its per-GPU SPP: the configuration the committed profiles describe),
        "config4" (same geometry, 1000 SPP in all) or False (custom sizes)."""
        self.rt, self.torch, self.dist = rt, torch, dist
        self.wl_name, self.mode, self.m = wl_name, mode, MODES[mode]
        self.W, self.H, self.spp, self.n_pix = W, H, spp, W * H
        self.launcher, self.rank, self.world, self.n_gpus, self.backend, self.film = launcher, rank, world, n_gpus, backend, film
        self.workload_id = (wl_name if full_size is True else f"{wl_name}-{full_size}" if full_size else f"{wl_name}-custom") + ("-dev" if device_bvh else "") + self.m["suffix"]
        self.device = torch.device("cuda", local_rank)
        self.tuning, build_opts = tuning_from_env()
        t0 = time.time()
        dev_arg = list(range(n_gpus)) if launcher == "group" else local_rank
        self.dev = rt.DeviceScene(scene, device=dev_arg, wide=self.m["wide"], device_bvh=device_bvh, **build_opts)
        self.t_create = time.time() - t0
        self.build_times = self.dev.build_times()
        self.ranks_formed = self.dev.n_devices if launcher == "group" else world
        self.block = SHARD_ROWS * W
        sharding = importlib.import_module("raytracing-course-hw-public_amd.sharding")
        self.fb = torch.zeros(self.n_pix * 3, dtype=torch.float32, device=self.device)
        self.img = torch.zeros(self.n_pix * 3, dtype=torch.uint8, device=self.device) if film else self.fb
        self.gather_device = self.device if backend == "nccl" else torch.device("cpu")
        self.gather = sharding.FramebufferGather(self.n_pix, self.block, rank, world, self.gather_device, dtype=self.img.dtype, all_gather=all_gather)
        self.my_pixels = sharding.shard_pixels(self.n_pix, self.block, rank, world)
        torch.cuda.synchronize()  # RT_FLAG_DEVICE_FB precondition (rt_abi.h): the zero-fills above ran on torch's stream

    def shard_kw(self):
        # torchrun: this rank renders its blocks; group: the library shards over its own GPUs (shard_count must stay 1)
        if self.world > 1:
            return dict(self.tuning, shard_index=self.rank, shard_count=self.world, shard_block=self.block)
        return dict(self.tuning, shard_block=self.block if self.launcher == "group" else 0)

    def step(self):
        if self.film:
            _, st = self.dev.run_raytracer_rgb8(self.W, self.H, self.spp, seed=SEED, device_rgb8=self.img.data_ptr(), global_best=self.m["gbest"], **self.shard_kw())
        else:
            _, st = self.dev.run_raytracer(self.W, self.H, self.spp, seed=SEED, device_fb=self.fb.data_ptr(), global_best=self.m["gbest"], **self.shard_kw())
        # torchrun, N > 1: RCCL gather of this rank's interleaved blocks to rank 0 over xGMI (no-op at N = 1; inside the library for "group")
        self.gather.gather(self.img if self.backend == "nccl" else self.img.cpu())
        return st

    def sync(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, steps, warmup):
        for _ in range(warmup):
            self.step()
        self.sync()
        kernel_ms, dom_ms, dom_launches = [], 0.0, 0
        t0 = time.perf_counter()
        for _ in range(steps):
            st = self.step()
            kernel_ms.append(st["kernel_ms"])
            dom_ms += st["dominant_ms"]
            dom_launches += st["dominant_launches"]
        self.sync()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.gather_device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        self.elapsed, self.steps = elapsed, steps
        self.kernel_ms, self.dom_ms, self.dom_launches = kernel_ms, dom_ms, dom_launches
        self.last_stats = st
        return float(self.n_pix) * self.spp * steps / elapsed / 1e6

    def roofline(self, live=None):
        """Roofline record of the closest-hit kernel for the steps just timed. Event counters come from the instrumented kernel
        variant on a bounded budget (<= 64 M samples) and are scaled to the step's sample count. `live`: (dict, source) from live_pmc():
        HBM traffic and SQ figures measured by child passes of this run; they take precedence over the committed profile summaries."""
        cnt_spp = max(1, min(self.spp, (64 << 20) // max(1, self.my_pixels)))
        _, cst = self.dev.run_raytracer(self.W, self.H, cnt_spp, seed=SEED, device_fb=self.fb.data_ptr(), counters=True, global_best=self.m["gbest"], **self.shard_kw())
        scale = self.spp / cnt_spp
        pixels_counted = self.my_pixels if self.launcher != "group" else self.n_pix
        all_bytes = algorithmic_bytes(cst, 0) * scale + 12.0 * pixels_counted
        trav_bytes = float(cst["box_tests"] * 24 + cst["nodes_visited"] * 16 + cst["tri_tests"] * 36) * scale
        if self.launcher == "group":  # counters are summed over the GPUs, launches run concurrently: per-GPU share
            trav_bytes /= self.n_gpus
        launches = self.dom_launches / self.steps
        launch_s = self.dom_ms / max(1, self.dom_launches) / 1e3
        algorithmic = trav_bytes / launches / launch_s / 1e9
        kernel_s = (sum(self.kernel_ms) / len(self.kernel_ms)) / 1e3
        src_hash = kernel_source_hash()
        traffic = traffic_x1 = requests = request_roof = None
        traffic_source = "not collected: PMC counters cannot run inside the timed bench"
        pmc, limiter, binding = None, None, None
        if self.n_gpus == 1:
            tj, traffic_source = load_profile("hbm_traffic", self.workload_id, src_hash)
            if tj is not None:
                traffic, traffic_x1 = tj.get("hbm_bytes_per_launch"), tj.get("hbm_bytes_per_launch_fetch_x1")
                requests, request_roof = tj.get("read_requests_per_launch"), tj.get("request_roof_Greq_s")
                traffic_source = (f"{tj['_file']}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command on the same device "
                                  "sources (hash checked); FETCH_SIZE x2 + WRITE_SIZE per MI355X_MICROARCH.md; not measured in this run")
            pj, pmc_source = load_profile("pmc_wf_extend", self.workload_id, src_hash)
            if pj is not None:
                pmc = {k: pj.get(k) for k in ("valu_busy", "lanes_per_valu", "wait_any_frac", "l1_hit", "l2_hit", "salu_per_valu", "l1_accesses_per_clk_per_cu",
                                              "l1_roof_accesses_per_clk_per_cu", "l1_frac", "l1_miss_rate_Greq_s", "l1_pending_stall_frac", "shader_clock_ghz", "workload", "spp")}
                pmc["source"] = pj["_file"] + " (earlier rocprofv3 --pmc run, same device sources)"
                limiter = pj.get("limiter")
            else:
                pmc = {"source": None, "note": pmc_source}
        if live is not None and live[0] is not None:
            lv = live[0]
            traffic, traffic_x1, requests = lv["hbm_bytes_per_launch"], lv["hbm_bytes_per_launch_fetch_x1"], lv["read_requests_per_launch"]
            request_roof = request_roof or 55.0
            traffic_source = live[1]
            pmc = dict(pmc or {})
            for k in ("valu_busy", "lanes_per_valu", "wait_any_frac"):
                if lv.get(k) is not None:
                    pmc[k] = lv[k]
            pmc["live"] = {k: lv[k] for k in ("launches_counted", "avg_launch_ms_under_pmc", "seconds")}
        elif live is not None:
            traffic_source = f"{traffic_source}; live PMC: {live[1]}"
        hbm_rate = traffic / launch_s / 1e9 if traffic else None
        request_frac = requests / launch_s / 1e9 / request_roof if requests and request_roof else None
        l1_frac = pmc.get("l1_frac") if pmc else None
        cands = [c for c in (("vector-L1 tag access rate (profiles/r02_l1_roof.txt: 0.98 accesses per clock per CU)", l1_frac),
                             ("HBM request rate for random <= 64-byte gathers (profiles/r02_gather64_calibration.txt: 55 G requests/s)", request_frac),
                             ("HBM bandwidth (8 TB/s nominal)", hbm_rate / HBM_PEAK_GBS if hbm_rate else None)) if c[1]]
        if cands:
            name, frac = max(cands, key=lambda c: c[1])
            binding = {"name": name, "frac": round(frac, 4), "source": (pmc or {}).get("source") or traffic_source}
        stream_peak, stream_src = measured_stream_peak()
        kernel = ("wf_extend_wide<false> (8-wide quantised BVH, every bounce)" if self.m["wide"] else
                  f"wf_extend<false, {'true' if self.m['gbest'] else 'false'}> (closest hit; primary rays through wf_extend_packet while its packets stay coherent)")
        frac_x1 = round(traffic_x1 / launch_s / 1e9 / HBM_PEAK_GBS, 4) if traffic_x1 else None
        frac_x2 = round(hbm_rate / HBM_PEAK_GBS, 4) if hbm_rate else None
        return {
            "bound": "hbm",  # the roof the contract prices this path against (no dense contraction -> no MFMA roof)
            # the contract's formula: ALGORITHMIC bytes per launch / live launch time / peak. Cache hits are not subtracted, so on
            # a cache-resident working set it exceeds 1; the HBM-side fractions are right below, named
            "achieved": round(algorithmic, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(algorithmic / HBM_PEAK_GBS, 4),
            "frac_is": "frac = frac_algorithmic (SURVEY 8d: box tests x 24 + node visits x 16 + triangle tests x 36 bytes, cache hits included) / launch time / 8 TB/s: work done, > 1 when caches serve it. "
                       "HBM utilisation proper: frac_hbm_counters_x1 (FETCH_SIZE as counted: what this kernel's <= 64-byte gathers cost, tools/ubench/gather64.hip) and frac_hbm_counters_x2 (FETCH_SIZE doubled, the guide's gfx950 rule for streaming reads: an upper bound here)",
            "frac_algorithmic": round(algorithmic / HBM_PEAK_GBS, 4),
            "frac_hbm_counters_x1": frac_x1,
            "frac_hbm_counters_x2": frac_x2,
            "valu_busy": (pmc or {}).get("valu_busy"),
            "lanes_per_valu": (pmc or {}).get("lanes_per_valu"),
            "l1_pending_stall_frac": (pmc or {}).get("l1_pending_stall_frac"),
            "traffic": traffic,
            "traffic_fetch_x1": traffic_x1,
            "traffic_source": traffic_source,
            "frac_fetch_x1": frac_x1,
            "fetch_size_multiplier": "x2 is MI355X_MICROARCH.md's prescription (a streaming request moves a 128-B line); this kernel's gathers are <= 64-B records, for which "
                                     "tools/ubench/gather64.hip measures ONE request per record (profiles/r02_gather64_calibration.txt): x1 (frac_fetch_x1) is what the workload's own calibration supports, x2 an upper bound",
            "algorithmic_GBps": round(algorithmic, 2),
            "algorithmic_over_peak": round(algorithmic / HBM_PEAK_GBS, 4),
            "algorithmic_is": "SURVEY 8d: reference-layout record sizes x event counts (cache hits included) / live launch time; a statement about work done, it exceeds 1 when the working set is cache resident",
            "binding_roof": binding,
            "hbm_rate": round(hbm_rate, 2) if hbm_rate else None,
            "hbm_frac": round(hbm_rate / HBM_PEAK_GBS, 4) if hbm_rate else None,
            "request_rate_Greq_s": round(requests / launch_s / 1e9, 2) if requests else None,
            "request_roof_Greq_s": request_roof,
            "request_frac": round(request_frac, 4) if request_frac else None,
            "l1_frac": l1_frac,
            "limiter": limiter,
            "pmc": pmc,
            "peak_measured_read": stream_peak,
            "peak_measured_source": stream_src,
            "kernel": kernel + ": every closest-hit launch is timed",
            "kernel_src_sha16": src_hash,
            "launches_per_step": launches,
            "avg_launch_ms": round(launch_s * 1e3, 4),
            "algorithmic_bytes_per_launch": round(trav_bytes / launches, 1),
            "pipeline": {  # all kernels of one rt_render (generate, extend, shade, resolve) against all algorithmic bytes
                "algorithmic_GBps": round(all_bytes / kernel_s / 1e9, 2),
                "device_ms_per_step": round(kernel_s * 1e3, 3),
                "algorithmic_bytes_per_sample": round(all_bytes / (pixels_counted * self.spp), 1),
                "casts_per_sample": round(cst["casts"] / max(1, cst["samples"]), 3),
                "nodes_per_cast": round(cst["nodes_visited"] / max(1, cst["casts"]), 1),
                "tri_tests_per_cast": round(cst["tri_tests"] / max(1, cst["casts"]), 1),
            },
        }

    def float_image(self):
        """One more render of the timed configuration through rt_render: the linear float3 framebuffer as a device tensor (a copy)."""
        self.dev.run_raytracer(self.W, self.H, self.spp, seed=SEED, device_fb=self.fb.data_ptr(), global_best=self.m["gbest"], **self.shard_kw())
        return self.fb.clone()

    def compare_with(self, parity_fb):
        """The production contract at the depth the number is quoted at: this mode's full image against the parity-mode GPU image of the same
        run (itself pinned to the oracle by tests/test_gpu_parity.py). Pixels are compared bit for bit and against the 1e-5 relative band."""
        torch = self.torch
        mine = self.float_image().view(-1, 3)
        ref = parity_fb.view(-1, 3)
        bits = (mine.view(torch.int32) != ref.view(torch.int32)).any(dim=1)
        rel = ((mine - ref).abs() / ref.abs().clamp_min(1e-6)).max(dim=1).values
        beyond = rel > 1e-5
        worst = [int(i) for i in torch.nonzero(beyond).flatten()[:8].tolist()]
        return {"pixels": int(ref.shape[0]), "spp": self.spp, "pixels_differing_from_parity": int(bits.sum().item()), "pixels_beyond_1e-5_relative": int(beyond.sum().item()),
                "max_relative_difference": float(rel.max().item()), "first_pixels_beyond": worst,
                "compared": "linear float3 framebuffer of this mode vs the parity-mode framebuffer of the same invocation (same seed, same samples), on the device",
                "expected": "0 unless a path met an exact tie between two triangles, or a hit the reference's own near-local pruning skips (bvh.h:216-223: a far child whose rounded entry distance is >= the near hit "
                            "although a triangle inside rounds closer); tests/test_gpu_production.py and tests/test_gpu_s10m.py assert the same count at full size"}

    def packet_record(self):
        """What the primary-ray packet policy did in the last timed step (rt_stats: passes, packet passes, the kernel's own census)."""
        st = getattr(self, "last_stats", None) or {}
        lanes = st.get("packet_lanes_x100", 0) / 100.0
        return {"passes": st.get("passes"), "packet_passes": st.get("packet_passes"), "lanes_served_per_trip": lanes if lanes else None,
                "threshold": self.tuning.get("packet_min_lanes") or (20.0 if self.m["wide"] else 33.0), "mode": {None: "auto", 1: "off", 2: "on"}[self.tuning.get("packet_mode")],
                "policy": "primary rays go through the 64-ray packet kernel while its census (lanes served per trip, read back one bounce late) stays at or above the threshold for this image size / samples per pass"}

    def close(self):
        self.dev.close()


def cpu_baseline_leg(rt, scene, wl, W, H, cpu_seconds):
    """The oracle in reference-RNG + libm mode (the reference CPU path, raytracer.h:636-662) on every `share`-th 256-pixel
    span of the same image, all host threads the cgroup grants, SPP chosen from a short probe to land near --cpu-seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle

    t0 = time.time()
    orc = oracle.OracleScene(scene)
    t_build = time.time() - t0
    cores = effective_cores()
    share = wl["cpu_share"] if W * H >= 256 * wl["cpu_share"] * 4 else 16
    _, p = orc.run_raytracer(W, H, 1, rng_mode=rt.RT_RNG_REFERENCE, shard_index=0, shard_count=share, shard_block=256, threads=cores)
    rate = p["samples"] / (p["total_ms"] / 1e3)
    cpu_spp = int(max(1, min(64, round(cpu_seconds * rate / p["samples"]))))
    _, c = orc.run_raytracer(W, H, cpu_spp, rng_mode=rt.RT_RNG_REFERENCE, shard_index=0, shard_count=share, shard_block=256, threads=cores)
    orc.close()
    return {
        "value": round(c["samples"] / (c["total_ms"] / 1e3) / 1e6, 4),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"same scene, every {share}th 256-pixel span of the {W}x{H} image at {cpu_spp} SPP = {c['samples']} samples, {c['total_ms'] / 1e3:.1f} s, "
                   f"reference RNG + libm (BVH build {t_build:.1f} s excluded, as the GPU's is); port vs reference binary: profiles/r03_cpu_baseline_crosscheck.json"),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: sponza, with S-10M as extra_workloads.s10m on one GPU")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"], help="weak (= auto): 64 SPP per GPU at every N; strong = BASELINE config 4 (1000 SPP in all) as the headline")
    ap.add_argument("--live-pmc", default="auto", choices=["auto", "off"], help="auto: on one GPU, at a workload's full size, measure the headline kernel's HBM traffic and SQ "
                    "figures with child rocprofv3 passes of this command after the timed steps (about a minute); off: quote the committed, hash-checked profiles only")
    ap.add_argument("--no-config4", action="store_true", help="skip the extra config-4 record (one 1000-SPP step after the headline)")
    ap.add_argument("--gather", default="gather", choices=["gather", "allgather"], help="torchrun flow: dist.gather to rank 0, or all_gather_into_tensor")
    ap.add_argument("--mode", default="parity", choices=sorted(MODES), help="traversal mode of the HEADLINE value (default parity; the others are reported under 'production')")
    ap.add_argument("--bvh", default="reference", choices=["reference", "device"],
                    help="binary builder: reference = host build in the reference's exact topology (the parity tree); device = built on the GPU (rt_bvh_device.hip: PLOC, or the radix-tree LBVH with RT_DEVICE_BUILDER=lbvh). "
                         "--mode wide collapses whichever is chosen")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel of the whole job (default: by --scaling)")
    ap.add_argument("--triangles", type=int, default=0)
    ap.add_argument("--tex-size", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline: no production-mode records, no extra workload")
    ap.add_argument("--film", default="device", choices=["device", "none"], help="device: rt_render_rgb8 (film on the GPU, rgb8 gathered); none: rt_render (float3 gathered)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the torchrun flow on fewer GPUs (all ranks on GPU 0, gather staged through host)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    args = ap.parse_args()
    wl_name = args.workload or "sponza"
    wl = WORKLOADS[wl_name]

    import numpy as np  # noqa: F401
    import torch  # imported BEFORE the HIP library so that one HIP runtime serves both (same soname)

    rt = importlib.import_module("raytracing-course-hw-public_amd")
    lib_stamp = rt.lib().rt_source_stamp().decode()
    if lib_stamp != kernel_source_hash() and not os.environ.get("RT_AMD_LIB"):  # (tuning variants are measured under their own name)
        raise SystemExit(f"librt_amd.so was built from other sources (stamp {lib_stamp}) than this tree ({kernel_source_hash()}): "
                         "run ./build.sh first; a bench line must describe the sources it is committed with")
    in_process = int(os.environ.get("WORLD_SIZE", "1")) <= 1 and args.gpus > 1
    launcher, rank, local_rank, world = resolve_launch(args.gpus, visible_devices=torch.cuda.device_count() if in_process else None)
    dist = None
    if launcher == "torchrun":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if args.backend == "gloo":
        local_rank = 0  # rehearsal: every rank renders on GPU 0
    torch.cuda.set_device(local_rank)
    film = args.film == "device"

    W, H = args.width or wl["width"], args.height or wl["height"]
    spp, scaling = resolve_spp(args.scaling, args.gpus, wl, args.spp)
    n_tri = args.triangles or wl["triangles"]
    tex_size = args.tex_size or wl["tex_size"]
    full_geometry = (W, H, n_tri, tex_size) == (wl["width"], wl["height"], wl["triangles"], wl["tex_size"])
    full_size = (True if spp == wl["spp_per_gpu"] * args.gpus else "config4" if (spp == CONFIG4_SPP and wl_name == "sponza") else False) if full_geometry else False

    t0 = time.time()
    scene = make_scene(rt, wl, n_tri, tex_size, W / H)
    t_gen = time.time() - t0
    common = dict(launcher=launcher, rank=rank, world=world, n_gpus=args.gpus, local_rank=local_rank, backend=args.backend, film=film, device_bvh=args.bvh == "device", all_gather=args.gather == "allgather")
    run = Runner(rt, torch, dist, scene, wl_name, args.mode, W, H, spp, full_size=full_size, **common)
    if launcher == "group" and run.ranks_formed != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the library formed {run.ranks_formed} rank(s)")
    value = run.timed(args.steps, args.warmup)
    live = None
    if args.live_pmc == "auto" and args.gpus == 1 and rank == 0 and full_size is True and not os.environ.get("RT_AMD_LIB"):
        child = ["--workload", wl_name, "--mode", args.mode, "--bvh", args.bvh, "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-config4", "--live-pmc", "off", "--film", args.film]
        live = live_pmc(child, args.mode)
    roofline = run.roofline(live)
    roofline["packet"] = run.packet_record()
    elapsed = run.elapsed
    single = args.gpus == 1 and rank == 0
    extras = single and not args.no_extras
    parity_fb = run.float_image() if (extras and args.mode == "parity") else None
    config4 = None
    if wl_name == "sponza" and full_geometry and scaling != "strong" and not args.no_config4 and args.mode == "parity":
        # BASELINE config 4 beside the weak-scaling headline: 1000 SPP in all over the GPUs taking part, one timed step (10^9 samples)
        keep = (run.spp, run.elapsed, run.steps, run.kernel_ms, run.dom_ms, run.dom_launches, run.last_stats)
        run.spp = max(1, min(CONFIG4_SPP, (128 << 20) // max(1, run.my_pixels)))  # untimed: one sample pass of config 4's size grows the path workspace to it
        run.step()
        run.spp = CONFIG4_SPP
        v4 = run.timed(1, 0)
        config4 = {"value": round(v4, 3), "unit": "Msamples/s", "scaling": "strong", "spp": CONFIG4_SPP, "steps": 1, "warmup": "one sample pass (workspace growth)", "ms_per_step": round(run.elapsed * 1e3, 3), "n_gpus": args.gpus,
                   "passes": run.last_stats.get("passes"), "packet_passes": run.last_stats.get("packet_passes"),
                   "config": f"BASELINE config 4: {wl['label']} {W}x{H}, {CONFIG4_SPP} SPP in all, sharded over {args.gpus} GPU(s), traversal {args.mode}"}
        run.spp, run.elapsed, run.steps, run.kernel_ms, run.dom_ms, run.dom_launches, run.last_stats = keep
    setup = {"scene_generation": round(t_gen, 2), "rt_create_bvh_upload": round(run.t_create, 2), "scene_bvh_build": round(run.build_times["build_ms"] / 1e3, 4)}
    run.close()

    x_steps, x_warm = max(1, min(args.steps, 5)), max(1, min(args.warmup, 2))

    def production_records(scene_, wl_name_, W_, H_, spp_, full_, parity_value, skip, steps_, warm_, parity_fb_):
        out = {}
        for mode in ("global", "wide"):
            if mode == skip:
                continue
            # the wide record is the FULL production build: binary tree (PLOC) and collapse on the device; global-best keeps the parity tree
            kw = dict(common, device_bvh=True) if mode == "wide" else common
            r = Runner(rt, torch, dist, scene_, wl_name_, mode, W_, H_, spp_, full_size=full_, **kw)
            v = r.timed(steps_, warm_)
            rf = r.roofline()
            rf["packet"] = r.packet_record()
            out[mode] = {"traversal": MODES[mode]["text"] + (", binary tree (PLOC) and collapse built on the device" if mode == "wide" else ""), "workload_id": r.workload_id, "value": round(v, 3), "unit": "Msamples/s", "steps": steps_, "warmup": warm_,
                         "ms_per_step": round(r.elapsed / steps_ * 1e3, 3), "vs_parity": round(v / parity_value, 3) if parity_value else None,
                         "rt_create_s": round(r.t_create, 2), "build_ms": {k: round(v, 2) for k, v in r.build_times.items()}, "roofline": rf,
                         "image_vs_parity": r.compare_with(parity_fb_) if parity_fb_ is not None else None}
            r.close()
        out["parity_of_these_modes"] = ("image_vs_parity: the full image at the quoted SPP against the parity-mode image, counted in this run; tests/test_gpu_production.py, against the CPU oracle: closest-hit t bit-equal on every ray (5 fixtures, S-sponza 60 000 rays, S-10M 100 000 rays), "
                                        "index differences only on exact ties, the S-sponza 1000x1000x1 SPP framebuffer bit-identical to the oracle's; the wide tree may find a hit "
                                        "1 ulp CLOSER than the reference where the reference's own pruning skips it (coplanar overlapping triangles): counted there")
        return out

    production = None
    if extras and args.mode == "parity":
        production = production_records(scene, wl_name, W, H, spp, full_size, value, None, x_steps, x_warm, parity_fb)
        del parity_fb

    cpu_baseline = None
    if single and not args.no_cpu_baseline:
        cpu_baseline = cpu_baseline_leg(rt, scene, wl, W, H, args.cpu_seconds)

    extra_workloads = None
    if extras and args.workload is None and args.mode == "parity" and not (args.width or args.height or args.triangles or args.tex_size or args.spp):
        del scene
        xw = WORKLOADS["s10m"]
        t0 = time.time()
        xscene = make_scene(rt, xw, xw["triangles"], xw["tex_size"], 1.0)
        xt_gen = time.time() - t0
        xs, xk = max(1, min(args.steps, 3)), 1
        xr = Runner(rt, torch, dist, xscene, "s10m", "parity", xw["width"], xw["height"], xw["spp_per_gpu"], full_size=True, **common)
        xv = xr.timed(xs, xk)
        xrf = xr.roofline()
        xrf["packet"] = xr.packet_record()
        x_parity_fb = xr.float_image()
        rec = {"metric": xw["metric"], "value": round(xv, 3), "unit": "Msamples/s", "n_gpus": 1, "steps": xs, "warmup": xk, "ms_per_step": round(xr.elapsed / xs * 1e3, 3),
               "config": {"workload": f"S-10M synthetic (BVH cache stress, BASELINE config 5's scene): {xw['triangles']}+28 triangles, 16x3 {xw['tex_size']}^2 RGBA8 textures, "
                                      f"{xw['width']}x{xw['height']}, {xw['spp_per_gpu']} SPP, ray_depth 8, device RNG", "workload_id": "s10m", "traversal": MODES["parity"]["text"]},
               "roofline": xrf, "cpu_baseline": None,
               "cpu_baseline_skipped": "the oracle's reference-topology build of 10^7 triangles takes ~25 s on this box before a single sample: python bench.py --workload s10m times it",
               "setup_s": {"scene_generation": round(xt_gen, 2), "rt_create_bvh_upload": round(xr.t_create, 2), "scene_bvh_build": round(xr.build_times["build_ms"] / 1e3, 4)}}
        xr.close()
        rec["production"] = production_records(xscene, "s10m", xw["width"], xw["height"], xw["spp_per_gpu"], True, xv, "global", xs, xk, x_parity_fb)
        extra_workloads = {"s10m": rec}

    if rank == 0:
        shard_text = "single GPU"
        if launcher == "torchrun":
            shard_text = (f"interleaved {SHARD_ROWS}-row tiles over {world} GPU(s), one process per GPU (torch.distributed.run), "
                          f"{'RCCL' if args.backend == 'nccl' else 'gloo (rehearsal, staged through the host)'} gather of the {'rgb8 image' if film else 'float3 framebuffer'}")
        elif launcher == "group":
            shard_text = (f"interleaved {SHARD_ROWS}-row tiles over {args.gpus} GPU(s), ONE process: replicas + ncclCommInitAll inside rt_create_on, "
                          f"grouped ncclSend/ncclRecv gather of the {'rgb8 image' if film else 'float3 framebuffer'} (csrc/rt_group.cpp)")
        out = {
            "metric": wl["metric"],
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{wl['label']} synthetic ({'Sponza-sized' if wl_name == 'sponza' else 'BVH cache stress'}): {n_tri}+28 triangles, 16x3 {tex_size}^2 RGBA8 textures, {W}x{H}, {spp} SPP, ray_depth 8, device RNG",
                "workload_id": run.workload_id,
                "baseline_config": ("config 4 (1000 SPP over N GPUs)" if (scaling == "strong" and wl_name == "sponza") else "config 3 shape (64 SPP per GPU)" if wl_name == "sponza" else "config 5 shape (32 SPP per GPU)"),
                "width": W,
                "height": H,
                "spp": spp,
                "triangles": int(n_tri + 28),
                "sharding": shard_text,
                "launcher": launcher,
                "ranks_formed": run.ranks_formed,
                "film": "device (rt_render_rgb8)" if film else "none (linear float3)",
                "traversal": MODES[args.mode]["text"],
                "bvh": "reference topology, host build" if args.bvh == "reference" else "built on the device (PLOC; identical closest hits, different topology and counters)",
            },
            "roofline": roofline,
            "config4": config4,
            "cpu_baseline": cpu_baseline,
            "production": production,
            "extra_workloads": extra_workloads,
            "setup_s": setup,
        }
        print(json.dumps(out), flush=True)
    if launcher == "torchrun":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

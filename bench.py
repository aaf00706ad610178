#!/usr/bin/env python3
"""bench.py — headline benchmark of the render-loop hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload sponza|s10m]

Metric (BASELINE.json): Msamples/s, whole job.

Workloads (the reference ships no Sponza asset, so both are the deterministic synthetic sets it names):
  sponza (default, BASELINE config 3/4): "S-sponza" = 262 144 random triangles in a closed 40x16x20 room + 12 wall
         triangles + 16 emissive ceiling triangles, 16 procedural 1024^2 RGBA8 texture sets, 66 materials, white
         environment, ray_depth 8; 1000x1000, 64 SPP per GPU.
  s10m   (BASELINE config 5): "S-10M" = the same recipe with 10^7 triangles (vertex offsets +-0.03), 2048x2048,
         32 SPP per GPU (256 SPP on 8 GPUs). Working set (nodes 0.64 GB + triangles 0.48 GB + attributes 1.28 GB) is far
         beyond the 256 MiB Infinity Cache: the configuration where HBM traffic / time / 8 TB/s is a real fraction.

One step = one pass of the hot path over one batch = one full render of the image: rt_render_rgb8() through the C-ABI
(render + the film on the device, i.e. what run_raytracer leaves in the reference's Image) with the image resident in
HBM (RT_FLAG_DEVICE_FB) + for N > 1 the RCCL gather of the rgb8 image to rank 0 (--film none: rt_render() and the
float3 framebuffer instead). Scene upload and BVH build happen once before the timed region, like the reference's
RaytracerStaticContext (raytracer.h:633) precedes its pixel loop.

N > 1 (launched by torch.distributed.run, one rank per GPU): the image is sharded in interleaved 8-row tiles
(block b -> rank b % N), scene replicated per GPU; weak scaling: SPP = spp_per_gpu * N so per-GPU work is fixed.

The "roofline" object (dominant kernel wf_extend):
  achieved / frac      SURVEY 8d's ALGORITHMIC bytes (reference-layout record sizes x event counts from the instrumented
                       kernel variant; cache hits are NOT subtracted) / the live HIP-event launch duration, against the
                       nominal 8 TB/s. This is the contract's figure; it can exceed 1 when the working set is cache
                       resident (S-sponza), so it must not be read as HBM utilisation.
  traffic / hbm_frac   HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE) kept
                       under profiles/ (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) and stamped with the
                       hash of the device sources they were measured on: used only
                       while that hash matches the sources of this run, else null. hbm_frac = traffic / live launch
                       duration / 8 TB/s is the fraction of the memory roofline actually used. request_frac = L2 read
                       requests per launch / live launch duration / the ~55 G requests/s this chip sustains for random
                       gathers of <= 64-B records (tools/ubench/gather64.hip, profiles/r02_gather64_calibration.txt): the
                       practical roof of this kernel's access pattern, which is request-rate bound, not byte bound.
  l1_frac              vector-L1 (TCP) tag accesses per clock per CU of wf_extend (same committed PMC profile) / the 0.98 the
                       chip retires at most (tools/l1_roof_probe.sh, profiles/r02_l1_roof.txt). A 64-byte node costs four
                       16-byte loads = four L1 accesses per lane whatever its cache residency: this is the roof that binds
                       the kernel on both workloads (S-sponza 0.8 of it at HBM 0.2; S-10M 0.7 with 42 % miss stalls).
  limiter + pmc        what the SQ/TCP/TCC counters of the same committed profile say binds the kernel (L1 access rate,
                       VALU issue share, active lanes per VALU instruction, wave-wait share, hit rates), with its source file.
"cpu_baseline": the CPU oracle (port of the reference algorithm, byte-identical to the reference binary on the
fixtures) timed on this box's host cores on a bounded sample of the same workload; rank 0, N = 1 only.
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
PROFILE_ROUND = "r02"

WORKLOADS = {
    "sponza": dict(label="S-sponza", width=1000, height=1000, spp_per_gpu=64, triangles=262144, offset=0.15, tex_size=1024, cpu_share=16,
                   metric="Msamples/sec (whole node) on Sponza 1000x1000"),
    "s10m": dict(label="S-10M", width=2048, height=2048, spp_per_gpu=32, triangles=10_000_000, offset=0.03, tex_size=1024, cpu_share=256,
                 metric="Msamples/sec (whole node) on synthetic 10M-triangle scene 2048x2048"),
}
DEVICE_SOURCES = ("raytracing-course-hw-public_amd/csrc/rt_wavefront.hip", "raytracing-course-hw-public_amd/csrc/rt_device_lib.h",
                  "raytracing-course-hw-public_amd/csrc/rt_device_types.h", "include/rt_devspec.h")


def kernel_source_hash() -> str:
    """sha256 over the device sources of the wavefront kernels: stamps PMC profiles so that a stale one is never quoted."""
    h = hashlib.sha256()
    for rel in DEVICE_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
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


def measured_stream_peak():
    """Best streaming-read rate of the box class from the committed microbenchmark record (nominal peak stays 8 TB/s)."""
    for rnd in (PROFILE_ROUND, "r01"):
        rel = os.path.join("profiles", f"{rnd}_hbm_stream.txt")
        try:
            vals = [float(v) for v in re.findall(r"read\s+([0-9.]+)\s*GB/s", open(os.path.join(ROOT, rel)).read())]
            if vals:
                return max(vals), rel
        except OSError:
            continue
    return None, None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="sponza", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (default: the workload's SPP per GPU x gpus)")
    ap.add_argument("--triangles", type=int, default=0)
    ap.add_argument("--tex-size", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--film", default="device", choices=["device", "none"], help="device: rt_render_rgb8 (film on the GPU, rgb8 gathered); none: rt_render (float3 gathered)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the N>1 flow on fewer GPUs (all ranks on GPU 0, gather staged through host)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--bvh", default="reference", choices=["reference", "device"],
                    help="reference: host build in the reference's exact topology (parity mode, the headline); device: LBVH built on the GPU (production mode: same closest hits, other topology)")
    ap.add_argument("--wide", action="store_true", help="production build: collapse the scene BVH into the 8-wide quantised tree (RT_BUILD_WIDE) and walk that")
    ap.add_argument("--traversal", default="reference", choices=["reference", "global"],
                    help="reference: the reference's traversal order and pruning (parity mode, the headline); global: prune against the global best hit (production: fewer node visits, same hits)")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    gbest = args.traversal == "global"

    import numpy as np  # noqa: F401
    import torch  # imported BEFORE the HIP library so that one HIP runtime serves both (same soname)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if args.backend == "gloo":
        local_rank = 0  # rehearsal: every rank renders on GPU 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    rt = importlib.import_module("raytracing-course-hw-public_amd")
    W, H = args.width or wl["width"], args.height or wl["height"]
    spp = args.spp if args.spp > 0 else wl["spp_per_gpu"] * world
    n_tri = args.triangles or wl["triangles"]
    tex_size = args.tex_size or wl["tex_size"]
    n_pix = W * H
    full_size = (W, H, n_tri, tex_size) == (wl["width"], wl["height"], wl["triangles"], wl["tex_size"]) and spp == wl["spp_per_gpu"] * world
    workload_id = (args.workload if full_size else f"{args.workload}-custom") + ("" if args.bvh == "reference" else "-lbvh") + ("-gbest" if gbest else "") + ("-wide" if args.wide else "")

    t0 = time.time()
    scene = rt.scenegen.room_scene(n_tri, seed=SEED, tex_size=tex_size, n_tex_sets=16, n_materials=64, n_lights=16,
                                   light_strength=20.0, alpha_fraction=0.02, offset=wl["offset"],
                                   camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9, aspect=W / H))
    t_gen = time.time() - t0
    t0 = time.time()
    dev = rt.DeviceScene(scene, device=local_rank, device_bvh=args.bvh == "device", wide=args.wide)
    t_create = time.time() - t0
    build_times = dev.build_times()

    block = SHARD_ROWS * W
    sharding = importlib.import_module("raytracing-course-hw-public_amd.sharding")
    fb = torch.zeros(n_pix * 3, dtype=torch.float32, device=device)
    film = args.film == "device"
    img = torch.zeros(n_pix * 3, dtype=torch.uint8, device=device) if film else fb
    gather_device = device if args.backend == "nccl" else torch.device("cpu")
    gather = sharding.FramebufferGather(n_pix, block, rank, world, gather_device, dtype=img.dtype)
    my_pixels = sharding.shard_pixels(n_pix, block, rank, world)

    def step():
        if film:
            _, st = dev.run_raytracer_rgb8(W, H, spp, seed=SEED, shard_index=rank, shard_count=world, shard_block=block, device_rgb8=img.data_ptr(), global_best=gbest)
        else:
            _, st = dev.run_raytracer(W, H, spp, seed=SEED, shard_index=rank, shard_count=world, shard_block=block, device_fb=fb.data_ptr(), global_best=gbest)
        # N > 1: RCCL gather of this rank's interleaved blocks to rank 0 over xGMI (no-op at N = 1)
        gather.gather(img if args.backend == "nccl" else img.cpu())
        return st

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()  # RT_FLAG_DEVICE_FB precondition (rt_abi.h): the zero-fills above ran on torch's stream
    for _ in range(args.warmup):
        step()
    sync()
    kernel_ms, dom_ms, dom_launches = [], 0.0, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = step()
        kernel_ms.append(st["kernel_ms"])
        dom_ms += st["dominant_ms"]
        dom_launches += st["dominant_launches"]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_samples = float(n_pix) * spp * args.steps
    value = total_samples / elapsed / 1e6

    # ---- roofline of the dominant kernel, wf_extend (closest-hit traversal): algorithmic bytes per launch / average
    # launch duration. Its algorithmic bytes are the scene-BVH traversal terms of SURVEY 8d (box tests x 24 + nodes x 16 +
    # triangle tests x 36); one render = `dominant_launches` launches (passes x bounces). Counters are per-sample event
    # counts; they are collected on a bounded budget (<= 64 M samples) and scaled to the step's sample count.
    cnt_spp = max(1, min(spp, (64 << 20) // max(1, my_pixels)))
    _, cst = dev.run_raytracer(W, H, cnt_spp, seed=SEED, shard_index=rank, shard_count=world, shard_block=block, device_fb=fb.data_ptr(), counters=True, global_best=gbest)
    scale = spp / cnt_spp
    all_bytes = algorithmic_bytes(cst, 0) * scale + 12.0 * my_pixels
    trav_bytes = float(cst["box_tests"] * 24 + cst["nodes_visited"] * 16 + cst["tri_tests"] * 36) * scale
    launches_per_render = dom_launches / args.steps
    avg_launch_s = dom_ms / max(1, dom_launches) / 1e3
    achieved = trav_bytes / launches_per_render / avg_launch_s / 1e9
    avg_kernel_s = (sum(kernel_ms) / len(kernel_ms)) / 1e3

    src_hash = kernel_source_hash()
    traffic, traffic_source = None, "not collected: PMC counters cannot run inside the timed bench"
    pmc, limiter, requests, request_roof = None, None, None, None
    if world == 1:
        tj, traffic_source = load_profile("hbm_traffic", workload_id, src_hash)
        if tj is not None:
            traffic = tj.get("hbm_bytes_per_launch")
            requests, request_roof = tj.get("read_requests_per_launch"), tj.get("request_roof_Greq_s")
            traffic_source = (f"{tj['_file']}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command on the same device "
                              "sources (hash checked); FETCH_SIZE x2 + WRITE_SIZE per MI355X_MICROARCH.md; not measured in this run")
        pj, pmc_source = load_profile("pmc_wf_extend", workload_id, src_hash)
        if pj is not None:
            pmc = {k: pj.get(k) for k in ("valu_busy", "lanes_per_valu", "wait_any_frac", "l1_hit", "l2_hit", "salu_per_valu", "l1_accesses_per_clk_per_cu",
                                          "l1_roof_accesses_per_clk_per_cu", "l1_frac", "l1_miss_rate_Greq_s", "l1_pending_stall_frac", "shader_clock_ghz", "workload", "spp")}
            pmc["source"] = pj["_file"] + " (earlier rocprofv3 --pmc run, same device sources)"
            limiter = pj.get("limiter")
        else:
            pmc = {"source": None, "note": pmc_source}
    stream_peak, stream_src = measured_stream_peak()
    roofline = {
        "bound": "hbm",  # the roof the contract prices this path against (no dense contraction -> no MFMA roof)
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "achieved_is": "ALGORITHMIC bytes (SURVEY 8d: reference-layout record sizes x event counts, cache hits included) / live launch time; NOT HBM utilisation: see hbm_frac",
        "traffic": traffic,
        "traffic_source": traffic_source,
        "hbm_rate": round(traffic / avg_launch_s / 1e9, 2) if traffic else None,
        "hbm_frac": round(traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
        "request_rate_Greq_s": round(requests / avg_launch_s / 1e9, 2) if requests else None,
        "request_roof_Greq_s": request_roof,
        "request_frac": round(requests / avg_launch_s / 1e9 / request_roof, 4) if requests and request_roof else None,
        "l1_frac": pmc.get("l1_frac") if pmc else None,  # vector-L1 tag accesses per clock per CU / the measured 0.98 roof (profiles/r02_l1_roof.txt)
        "limiter": limiter,
        "pmc": pmc,
        "peak_measured_read": stream_peak,
        "peak_measured_source": stream_src,
        "kernel": "wf_extend<false> (closest hit; primary rays through wf_extend_packet<false> while its packets stay coherent): every closest-hit launch is timed",
        "kernel_src_sha16": src_hash,
        "launches_per_step": launches_per_render,
        "avg_launch_ms": round(avg_launch_s * 1e3, 4),
        "algorithmic_bytes_per_launch": round(trav_bytes / launches_per_render, 1),
        "pipeline": {  # all kernels of one rt_render (generate, extend, shade, resolve) against all algorithmic bytes
            "achieved": round(all_bytes / avg_kernel_s / 1e9, 2),
            "frac": round(all_bytes / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 4),
            "device_ms_per_step": round(avg_kernel_s * 1e3, 3),
            "algorithmic_bytes_per_sample": round(all_bytes / (my_pixels * spp), 1),
            "casts_per_sample": round(cst["casts"] / max(1, cst["samples"]), 3),
            "nodes_per_cast": round(cst["nodes_visited"] / max(1, cst["casts"]), 1),
            "tri_tests_per_cast": round(cst["tri_tests"] / max(1, cst["casts"]), 1),
        },
    }

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle

        t0 = time.time()
        orc = oracle.OracleScene(scene)
        t_oracle_build = time.time() - t0
        cores = effective_cores()
        # bounded sample of the same workload: every `share`-th 256-pixel span of the same image, reference RNG + libm
        # (the reference CPU path, raytracer.h:636-662), SPP chosen from a short probe to land near --cpu-seconds.
        share = wl["cpu_share"] if n_pix >= 256 * wl["cpu_share"] * 4 else 16
        _, p = orc.run_raytracer(W, H, 1, rng_mode=rt.RT_RNG_REFERENCE, shard_index=0, shard_count=share, shard_block=256, threads=cores)
        rate = p["samples"] / (p["total_ms"] / 1e3)
        cpu_spp = int(max(1, min(64, round(args.cpu_seconds * rate / p["samples"]))))
        _, c = orc.run_raytracer(W, H, cpu_spp, rng_mode=rt.RT_RNG_REFERENCE, shard_index=0, shard_count=share, shard_block=256, threads=cores)
        cpu_baseline = {
            "value": round(c["samples"] / (c["total_ms"] / 1e3) / 1e6, 4),
            "unit": "Msamples/s",
            "cores": cores,
            "kind": "port",
            "sample": (f"same scene, every {share}th 256-pixel span of the {W}x{H} image at {cpu_spp} SPP = {c['samples']} samples, {c['total_ms'] / 1e3:.1f} s, "
                       f"reference RNG + libm (BVH build {t_oracle_build:.1f} s excluded, as the GPU's is)"),
        }
        orc.close()

    if rank == 0:
        out = {
            "metric": wl["metric"],
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{wl['label']} synthetic ({'Sponza-sized' if args.workload == 'sponza' else 'BVH cache stress'}): {n_tri}+28 triangles, 16x3 {tex_size}^2 RGBA8 textures, {W}x{H}, {spp} SPP, ray_depth 8, device RNG",
                "workload_id": workload_id,
                "width": W,
                "height": H,
                "spp": spp,
                "triangles": int(scene.n_triangles),
                "sharding": f"interleaved {SHARD_ROWS}-row tiles over {world} GPU(s), RCCL gather of the {'rgb8 image' if film else 'float3 framebuffer'}" if world > 1 else "single GPU",
                "film": "device (rt_render_rgb8)" if film else "none (linear float3)",
                "traversal": "8-wide quantised BVH, global-best culling, octant order (production)" if args.wide else "global-best pruning (production)" if gbest else "reference order and pruning (parity mode)",
                "bvh": "reference topology, host build (parity mode)" if args.bvh == "reference" else "LBVH built on the device (production mode: identical closest hits, different topology and counters)",
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "setup_s": {"scene_generation": round(t_gen, 2), "rt_create_bvh_upload": round(t_create, 2), "scene_bvh_build": round(build_times["build_ms"] / 1e3, 4)},
        }
        print(json.dumps(out), flush=True)
    dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

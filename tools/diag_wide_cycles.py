#!/usr/bin/env python3
"""tools/diag_wide_cycles.py [sponza|s10m] [spp] — development aid: section census of wf_extend_wide (-DRT_DIAG_CYCLES variant 'wcyc': wave cycles
between s_memtime stamps for refill / unwind / node step / triangle batch, trips and lanes per kind of trip) over the secondary bounces of one render
of the bench scene with the production build (PLOC + collapse on the device). Primary rays go through the per-lane kernel too (packets off)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "raytracing-course-hw-public_amd/csrc/variants/%s.so" % os.environ.get("RT_DIAG_VARIANT", "wcyc"))
import torch  # noqa: F401  (one HIP runtime)
import bench
rt = importlib.import_module("raytracing-course-hw-public_amd")
wl_name = sys.argv[1] if len(sys.argv) > 1 else "s10m"
wl = bench.WORKLOADS[wl_name]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = bench.make_scene(rt, wl, wl["triangles"], 64, wl["width"] / wl["height"])
dev = rt.DeviceScene(sc, device_bvh=True, wide=True)
out = np.zeros(32, dtype=np.uint64)
lib = rt.lib(); lib.rt_debug_census.argtypes = [C.c_void_p, C.c_void_p]
W, H = wl["width"], wl["height"]
dev.run_raytracer(W, H, spp, seed=1, packet_mode=rt.RT_PACKET_OFF)
lib.rt_debug_census(dev._h, out.ctypes.data)
_, st = dev.run_raytracer(W, H, spp, seed=1, packet_mode=rt.RT_PACKET_OFF)
lib.rt_debug_census(dev._h, out.ctypes.data)
o = [float(x) for x in out]
tot = o[4]
print(f"wf_extend_wide section census, {wl['label']} {W}x{H}x{spp} (production build, packets off): kernel_ms {st['kernel_ms']:.1f}, closest-hit launches {st['dominant_launches']} / {st['dominant_ms']:.1f} ms")
for i, nm in enumerate(["refill + loop head", "unwind (pop)", "node steps", "triangle batches"]):
    print(f"  {nm:20s} {o[i] / tot * 100:5.1f} % of wave cycles")
print(f"  node-step trips {int(o[6])}: {o[7] / max(1, o[6]):.1f} lanes stepping, {o[2] / max(1, o[6]):.0f} wave cycles per trip")
print(f"  triangle-batch trips {int(o[8])}: {o[9] / max(1, o[8]):.1f} lanes waiting, {o[3] / max(1, o[8]):.0f} wave cycles per trip")
print(f"  waves {int(o[5])}, wave cycles per wave {tot / max(1, o[5]):.0f}")

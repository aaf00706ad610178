#!/usr/bin/env python3
"""tools/small_render_timing.py — wall time of small renders (BASELINE config 1's shape is 256 x 256 x 4): the S-sponza scene at reduced
image sizes in parity and production mode, and tests/golden/txt/scene-000.txt at 256 x 256 x 4. Best of 20 after 3 warm-up renders."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa: F401
import bench
rt = importlib.import_module("raytracing-course-hw-public_amd")
wl = bench.WORKLOADS["sponza"]
sc = bench.make_scene(rt, wl, wl["triangles"], 256, 1.0)
def best(dev, W, H, spp, **kw):
    for _ in range(3):
        dev.run_raytracer(W, H, spp, seed=3, **kw)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); _, st = dev.run_raytracer(W, H, spp, seed=3, **kw); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, st["kernel_ms"]
for name, kw in (("parity", {}), ("production", dict(device_bvh=True, wide=True))):
    dev = rt.DeviceScene(sc, **kw)
    for W, H, spp in ((64, 64, 8), (256, 256, 4), (1000, 1000, 1), (1000, 1000, 4)):
        for label, tk in (("auto", {}), ("sorted (octant+cell+cone)", dict(sort_mode=rt.RT_SORT_OCTANT_CELL_CONE))):
            wall, dev_ms = best(dev, W, H, spp, **tk)
            print(f"S-sponza {name:10s} {W}x{H}x{spp}: {label:26s} wall {wall:6.2f} ms, device {dev_ms:6.2f} ms, {W * H * spp / wall / 1e3:7.1f} Msamples/s")
    dev.close()
ls = rt.parse_scene_txt(os.path.join(ROOT, "tests", "golden", "txt", "scene-000.txt"))
dev = rt.DeviceScene(ls)
for label, tk in (("auto", {}), ("sorted (octant+cell+cone)", dict(sort_mode=rt.RT_SORT_OCTANT_CELL_CONE))):
    wall, dev_ms = best(dev, 256, 256, 4, **tk)
    print(f"config 1 (scene-000.txt) 256x256x4: {label:26s} wall {wall:6.2f} ms, device {dev_ms:6.2f} ms, {256 * 256 * 4 / wall / 1e3:7.1f} Msamples/s")

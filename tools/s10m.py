#!/usr/bin/env python3
"""tools_s10m.py — development aid: BASELINE config 5 shape (10M random triangles, 2048x2048) smoke/timing on one GPU."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracing-course-hw-public_amd")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t = time.time()
sc = rt.scenegen.room_scene(n, seed=0x5EED5EED, tex_size=256, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0, alpha_fraction=0.02,
                            offset=0.03, camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
print("scene gen s", round(time.time() - t, 1), flush=True)
t = time.time(); dev = rt.DeviceScene(sc); print("rt_create (2 BVH builds + upload) s", round(time.time() - t, 1), flush=True)
b = dev.bvh_info(0); print("nodes", len(b["nodes"]), "leaves", int((b["nodes"][:, 6] == 0xFFFFFFFF).sum()), flush=True)
fb, st = dev.run_raytracer(2048, 2048, spp, seed=1)
fb, st = dev.run_raytracer(2048, 2048, spp, seed=1)
print("render ms", st["kernel_ms"], "Msamples/s", 2048 * 2048 * spp / st["kernel_ms"] / 1e3, "finite", bool(np.isfinite(fb).all()), "mean", float(fb.mean()), flush=True)
_, c = dev.run_raytracer(2048, 2048, 1, seed=1, counters=True)
print("casts/sample", c["casts"] / c["samples"], "nodes/cast", c["nodes_visited"] / c["casts"], "tri/cast", c["tri_tests"] / c["casts"])

#!/usr/bin/env python3
"""tools/lbvh_probe.py <triangles> — render speed of the device LBVH against the reference-topology tree, by LBVH leaf size."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa: F401
rt = importlib.import_module("raytracing-course-hw-public_amd")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
W = H = 1024
sc = rt.scenegen.room_scene(n, seed=0x5EED5EED, tex_size=64, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0, alpha_fraction=0.02,
                            offset=0.03 if n >= 5_000_000 else 0.15, camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
def run(tag, **kw):
    dev = rt.DeviceScene(sc, **kw)
    dev.run_raytracer(W, H, 2, seed=1)
    _, st = dev.run_raytracer(W, H, 4, seed=1)
    _, c = dev.run_raytracer(W, H, 1, seed=1, counters=True)
    print(tag, dev.build_times(), "render 4 SPP ms", round(st["kernel_ms"], 1), "nodes/cast", round(c["nodes_visited"] / c["casts"], 1), "tri tests/cast", round(c["tri_tests"] / c["casts"], 1), flush=True)
    dev.close()
run("reference topology")
for leaf in (1, 2, 4, 8):
    os.environ["RT_LBVH_LEAF"] = str(leaf)
    run(f"device LBVH leaf {leaf}", device_bvh=True)

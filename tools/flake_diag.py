#!/usr/bin/env python3
"""tools/flake_diag.py — development aid: several scenes in one process, first render of each is a sorted wavefront render
(the pattern that exposed the null-stream memset race of ensure_wavefront)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
rt = importlib.import_module("raytracing-course-hw-public_amd")
import oracle
sc = rt.scenegen.room_scene(262144, seed=0x5EED5EED, tex_size=256, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0,
                            alpha_fraction=0.02, offset=0.15, camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
W = H = 1000
orc = oracle.OracleScene(sc)
o, so = orc.run_raytracer(W, H, 1, seed=0x5EED5EED)
def d(x, y): return int((x.view(np.uint32) != y.view(np.uint32)).any(axis=2).sum())
keep = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    dev = rt.DeviceScene(sc)
    c, _ = dev.run_raytracer(W, H, 1, seed=0x5EED5EED, counters=bool(i & 1))
    print("scene", i, "first render diff vs oracle", d(c, o), flush=True)
    if i % 3 == 0:
        keep.append(dev)
    else:
        dev.close()

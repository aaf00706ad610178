#!/usr/bin/env python3
"""tools/soak_bench_scene.py [n_views] [seed] [spp] — development aid: the benched production build against the parity image on the bench scene, from random viewpoints.

bench.py counts the pixels in which the production image (8-wide tree built on the device) differs from the parity image for ONE camera (0 at 64 SPP). This
script repeats the count for random cameras and seeds on S-sponza (262 144 triangles, 1000 x 1000, default 16 SPP), for the production build and for
global-best pruning, and has every differing pixel explained through the oracle (tests/conftest.py::explain_differing_pixels: an exact tie, or a closer hit the
reference's pruning skips — anything else raises)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("raytracing-course-hw-public_amd")
import oracle  # noqa: E402  (checker only)
from conftest import explain_differing_pixels  # noqa: E402


def main():
    n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    sg = rt.scenegen
    W = H = 1000
    t0 = time.time()
    total = {"wide": 0, "gbest": 0}
    for v in range(n_views):
        cam = sg.look_camera((float(rng.uniform(-18, 18)), float(rng.uniform(1, 15)), float(rng.uniform(-8, 8))), yaw_deg=float(rng.uniform(-180, 180)), yfov=float(rng.uniform(0.5, 1.3)))
        sc = sg.room_scene(262144, seed=0x5EED5EED, tex_size=256, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0, alpha_fraction=0.02, offset=0.15, camera=cam)
        seed = int(rng.integers(0, 2**31))
        par = rt.DeviceScene(sc)
        wide = rt.DeviceScene(sc, wide=True, device_bvh=True)
        orc = None
        try:
            pfb, _ = par.run_raytracer(W, H, spp, seed=seed)
            wfb, _ = wide.run_raytracer(W, H, spp, seed=seed)
            gfb, _ = par.run_raytracer(W, H, spp, seed=seed, global_best=True)
            msg = []
            for name, fb, dev in (("wide", wfb, wide), ("gbest", gfb, par)):
                bits = (fb.view(np.uint32) != pfb.view(np.uint32)).any(axis=2)
                n = int(bits.sum())
                total[name] += n
                msg.append(f"{name}: {n} of {W * H} pixels differ")
                if n and name == "wide":
                    orc = orc or oracle.OracleScene(sc)
                    for rec in explain_differing_pixels(rt, orc, par, dev, W, H, spp, seed, np.argwhere(bits)[:8]):
                        msg.append(str(rec))
            print(f"view {v}: camera {cam.position.tolist()} seed {seed}: " + "; ".join(msg), flush=True)
        finally:
            par.close()
            wide.close()
            if orc:
                orc.close()
    print(f"{n_views} views at {spp} SPP ({n_views * W * H * spp / 1e6:.0f} M samples per mode): production build {total['wide']} differing pixels, global best {total['gbest']}; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/tree_cost_probe.py [workload: sponza|s10m] — development aid: does a surface-area cost of the packed wide tree predict the node visits a render pays?

For a range of PLOC search radii (each gives another tree of the same scene: profiles/r04_variants.txt item 13) the production tree is built on the device, read back
(rt_bvh_wide_dump) and priced: sum over the slots of every node of (slot box area / root area) x (1 for an inner slot, triangles for a leaf slot) — the expected number of
node visits / triangle tests of a random long ray. Beside it: the node visits and triangle tests per cast of a real render (4 SPP, event counters), and the correlation."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("raytracing-course-hw-public_amd")
from test_wide_build import decode  # noqa: E402


def tree_cost(nodes):
    D = decode(nodes)
    cell = np.ldexp(1.0, D["e"] - 127)[:, :, None]  # (n, 3, 1)
    ext = np.maximum(D["qhi"] - D["qlo"], 0.0) * cell  # (n, 3, 8); empty slots are inverted: 0
    area = 2.0 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0])  # (n, 8)
    imask = D["imask"][:, None] >> np.arange(8)[None, :] & 1
    tbits = (D["tri_mask"][:, None] >> (3 * np.arange(8))[None, :]) & 7
    ntri = (tbits & 1) + ((tbits >> 1) & 1) + ((tbits >> 2) & 1)
    lo = (D["p"][0].astype(np.float64)[:, None] + D["qlo"][0] * cell[0])
    hi = (D["p"][0].astype(np.float64)[:, None] + D["qhi"][0] * cell[0])
    used = (imask[0] == 1) | (ntri[0] > 0)
    rlo, rhi = lo[:, used].min(axis=1), hi[:, used].max(axis=1)
    e = rhi - rlo
    root_area = 2.0 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0])
    return 1.0 + float((area * imask).sum() / root_area), float((area * ntri).sum() / root_area)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "sponza"
    sg = rt.scenegen
    if wl == "sponza":
        sc = sg.room_scene(262144, seed=0x5EED5EED, tex_size=64, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0, alpha_fraction=0.02, offset=0.15,
                           camera=sg.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
        W = H = 1000
    else:
        sc = sg.room_scene(10_000_000, seed=0x5EED5EED, tex_size=64, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0, alpha_fraction=0.02, offset=0.03,
                           camera=sg.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
        W = H = 1024
    rows = []
    for radius in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 24, 32):
        dev = rt.DeviceScene(sc, wide=True, device_bvh=True, ploc_radius=radius)
        try:
            dump = dev.bvh_wide_dump()
            cn, ct = tree_cost(dump["nodes"])
            _, st = dev.run_raytracer(W, H, 4, seed=1, counters=True)
            nv, tt = st["nodes_visited"] / st["casts"], st["tri_tests"] / st["casts"]
            rows.append((radius, cn, ct, nv, tt, len(dump["nodes"])))
            print(f"radius {radius:2d}: {len(dump['nodes']):8d} nodes; cost: {cn:7.2f} node visits + {ct:6.2f} triangle tests per unit of root area; render: {nv:6.1f} + {tt:5.1f} per cast", flush=True)
        finally:
            dev.close()
    r = np.array(rows)
    for wt in (0.0, 0.3, 1.0):
        print(f"correlation of (cost_nodes + {wt} cost_tris) with (visits + {wt} tests): {np.corrcoef(r[:, 1] + wt * r[:, 2], r[:, 3] + wt * r[:, 4])[0, 1]:.3f}")
    print(f"correlation of cost_nodes with visits per cast: {np.corrcoef(r[:, 1], r[:, 3])[0, 1]:.3f}")


if __name__ == "__main__":
    main()

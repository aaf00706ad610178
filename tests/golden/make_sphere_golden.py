#!/usr/bin/env python3
"""Known answers for the scene-txt ELLIPSOID's quadratic solve, from the reference (container only).

HEAD's render loop has no analytic primitive, but raytracer.h:61-77 still carries `intersect_ray_sphere(ray, r)` (unused). The ELLIPSOID of
include/rt_primspec.h restates exactly that solve (start / r, dir / r, a, hb, c, hd2, t1, t2) before it picks a root and builds a normal, so for a
sphere at the origin its distance must be one of the reference function's two roots, bit for bit. oracle/_ref/ref_probe calls the reference function;
tests/golden/sphere_kat.npz stores rays, radii and (t1, t2).    python tests/golden/make_sphere_golden.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402


def rays_for(seed, n, r):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3)) * r * 2.0
    o[: n // 4] *= 0.3  # origins inside the sphere
    tgt = rng.normal(size=(n, 3)) * r * 0.8
    d = tgt - o
    d[n // 2 :] = rng.normal(size=(n - n // 2, 3))  # half of them aimed anywhere: misses, grazing rays
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[-8:] *= 3.0  # direction need not be unit length
    return np.concatenate([o, d], axis=1).astype(np.float32)


def main():
    assert oracle.have_reference_build()
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for k, r in enumerate((1.0, 0.37, 25.0)):
            rays = rays_for(100 + k, 3000, r)
            rays.tofile(os.path.join(td, "rays.bin"))
            oracle.ref_probe("sphere", repr(float(np.float32(r))), 0, 0, os.path.join(td, "rays.bin"), os.path.join(td, "t.bin"))
            out[f"rays_{k}"] = rays
            out[f"radius_{k}"] = np.float32(r)
            out[f"t_{k}"] = np.fromfile(os.path.join(td, "t.bin"), dtype=np.float32).reshape(-1, 2)
    np.savez_compressed(os.path.join(HERE, "sphere_kat.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()}, "hits", [int((out[f"t_{k}"][:, 1] != 0).sum()) for k in range(3)])


if __name__ == "__main__":
    main()

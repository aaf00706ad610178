#!/usr/bin/env python3
"""tools/soak_prims.py [n_cases] [seed] — development aid: randomized soak of the scene-txt front end's analytic primitives on the GPU box.

Each case draws 1 - 10 ELLIPSOIDs (semi-axes log-uniform in [0.02, 20]: aspect ratios up to 1000, random rotations and positions) and 0 - 3 PLANEs, optionally
some BOX triangles around them, and checks through librt_amd.so:
  (a) hits and a small render against the CPU oracle, bit for bit (same include/rt_primspec.h on both sides: this pins scheduling and plumbing — the order in
      which wf_extend_prims and the BVH kernels see a ray, the strict-less rule between a triangle and a primitive, the production tree next to primitives);
  (b) hits against the float64 closed forms of tests/test_gpu_txt.py (written from the geometry, independent of that header): away from ill-conditioned
      answers the same hit / miss and primitive on every ray, t within 1500 float32 ulps of the length scale the solve works with.
Prints one line per case; exits 1 on the first violation."""
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
from test_gpu_txt import closed_form, length_scale  # noqa: E402  (the independent float64 solution)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    sg = rt.scenegen
    t0 = time.time()
    worst = 0.0
    for case in range(n_cases):
        def rq():
            q = rng.normal(size=4)
            return tuple(float(x) for x in (q / np.linalg.norm(q)).astype(np.float32))

        n_e, n_p = int(rng.integers(1, 11)), int(rng.integers(0, 4))
        prims = [dict(kind=1, material_id=int(rng.integers(0, 2)), param=tuple(float(x) for x in np.exp(rng.uniform(np.log(0.02), np.log(20.0), 3)).astype(np.float32)),
                      position=tuple(float(x) for x in rng.uniform(-10, 10, 3).astype(np.float32)), rotation=rq()) for _ in range(n_e)]
        prims += [dict(kind=2, material_id=0, param=tuple(float(x) for x in rng.normal(size=3).astype(np.float32)), position=tuple(float(x) for x in rng.uniform(-12, 12, 3).astype(np.float32)),
                       rotation=rq()) for _ in range(n_p)]
        n_t = int(rng.choice([0, 0, 12, 200]))
        pos = (rng.uniform(-10, 10, size=(n_t, 1, 3)) + rng.uniform(-1.5, 1.5, size=(n_t, 3, 3))).astype(np.float32)
        tang = np.tile(np.array([1, 0, 0], dtype=np.float32), (n_t, 3, 1))
        mats = [sg.Material(color=(0.8, 0.7, 0.6, 1.0), roughness=0.7, metallic=0.1), sg.Material(color=(1, 1, 1, 1), emission=(1.0, 0.9, 0.7), emissive_strength=4.0, roughness=1.0, metallic=0.0)]
        sc = sg.Scene(positions=pos, normals=None, texcoords=np.zeros((n_t, 3, 2), np.float32), tangents=tang, material_ids=np.zeros(n_t, np.uint32), materials=mats, textures=[],
                      camera=sg.look_camera((float(rng.uniform(-12, 12)), float(rng.uniform(-12, 12)), float(rng.uniform(12, 25))), yaw_deg=float(rng.uniform(-25, 25)), yfov=0.9), primitives=prims)
        n = 40_000
        o = rng.uniform(-14, 14, size=(n, 3))
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([o, d], axis=1).astype(np.float32)
        orc = oracle.OracleScene(sc)
        msgs = []
        try:
            op, ob = orc.cast_rays(rays)
            W, H, SPP = int(rng.integers(8, 90)), int(rng.integers(8, 70)), int(rng.choice([1, 4, 9]))
            ofb, ost = orc.run_raytracer(W, H, SPP, seed=case)
            for kw in ({}, dict(wide=True), dict(wide=True, device_bvh=True)) if n_t else ({},):
                dev = rt.DeviceScene(sc, **kw)
                try:
                    gp, gb = dev.cast_rays(rays)
                    if not kw:  # parity build: everything is the oracle's
                        gfb, gst = dev.run_raytracer(W, H, SPP, seed=case, counters=True)
                        if not (np.array_equal(gp, op) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32)) and np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32))
                                and gst["casts"] == ost["casts"] and gst["shaded_hits"] == ost["shaded_hits"]):
                            print(f"case {case}: parity build differs from the oracle: hits {int((gp != op).sum())} bct {int((gb.view(np.uint32) != ob.view(np.uint32)).any(axis=1).sum())} "
                                  f"pixels {int((gfb.view(np.uint32) != ofb.view(np.uint32)).any(axis=2).sum())}; prims {prims}", flush=True)
                            sys.exit(1)
                    else:  # production tree next to the primitives: the primitive hits are the oracle's wherever a primitive wins
                        prim_o = (op != 0xFFFFFFFF) & (op >= n_t)
                        same = prim_o & (gp == op)
                        if not (np.array_equal(gb[same].view(np.uint32), ob[same].view(np.uint32)) and not ((gp == 0xFFFFFFFF) & (op != 0xFFFFFFFF)).any()
                                and not (gb[(gp != 0xFFFFFFFF) & (op != 0xFFFFFFFF), 2] > ob[(gp != 0xFFFFFFFF) & (op != 0xFFFFFFFF), 2]).any()):
                            print(f"case {case}: {kw}: a primitive hit differs / a hit was lost / a farther hit; prims {prims}", flush=True)
                            sys.exit(1)
                    msgs.append("ok")
                finally:
                    dev.close()
            # (b) the independent float64 solution, primitives only (triangles would hide primitives: cast against a primitives-only copy)
            z = np.zeros((0, 3, 3), dtype=np.float32)
            sc_p = sg.Scene(positions=z, normals=z, texcoords=np.zeros((0, 3, 2), np.float32), tangents=z, material_ids=np.zeros(0, np.uint32), materials=mats, textures=[], camera=sc.camera, primitives=prims)
            dev = rt.DeviceScene(sc_p)
            try:
                gp, gb = dev.cast_rays(rays)
            finally:
                dev.close()
            ci, ct, _, margin = closed_form(prims, rays)
            good = margin > 1e-3
            got_i = np.where(gp == 0xFFFFFFFF, -1, gp.astype(np.int64))
            wrong = good & (got_i != ci)
            hit = good & (ci >= 0) & (got_i == ci)
            L = length_scale(prims, rays, ci)
            L = np.where(np.isfinite(ct), np.maximum(L, ct), L)
            err_ulps = np.abs(gb[hit, 2].astype(np.float64) - ct[hit]) / np.spacing(L[hit].astype(np.float32)).astype(np.float64)
            worst = max(worst, float(err_ulps.max(initial=0.0)))
            line = (f"case {case:3d}: {n_e} ellipsoids (aspect up to {max(max(p['param']) / min(p['param']) for p in prims[:n_e]):7.1f}) {n_p} planes {n_t} triangles: oracle {msgs}; float64: "
                    f"{int(good.sum())} well-conditioned rays, {int(wrong.sum())} with another answer, t error <= {float(err_ulps.max(initial=0.0)):.0f} ulps of the length scale")
            print(line, flush=True)
            if wrong.any() or err_ulps.max(initial=0.0) > 1500:
                i = int(np.flatnonzero(wrong)[0]) if wrong.any() else int(np.flatnonzero(hit)[np.argmax(err_ulps)])
                print(f"   ray {i}: {rays[i].tolist()} float64 prim {ci[i]} t {ct[i]} margin {margin[i]:.3e} | device prim {got_i[i]} t {gb[i, 2]}; prim {prims[int(ci[i])] if ci[i] >= 0 else None}", flush=True)
                sys.exit(1)
        finally:
            orc.close()
    print(f"{n_cases} cases: parity build == oracle in all; float64 closed form: same answer on every well-conditioned ray, worst t error {worst:.0f} ulps of the length scale; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

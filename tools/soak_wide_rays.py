#!/usr/bin/env python3
"""tools/soak_wide_rays.py [n_cases] [seed] — development aid: randomized soak of the production (8-wide) tree's closest hits on the GPU box.

The packed wide node keeps origins on ONE scene grid (20 bits per axis) and cell sizes as 4-bit exponents above a scene base (csrc/wide_grid.h), and the
traversal folds the ray-dependent half of every plane equation into per-ray constants built with exponent arithmetic (WRay, rt_wide.hip). This soak
stresses exactly that: random triangle soups scaled by 2^k (k in [-30, 40]) and moved far from the origin (offsets up to 2^12 extents: coordinates whose
ulp approaches the grid step), flat and needle-shaped scenes, and rays chosen to be awkward — axis-parallel, with zero and denormal-small direction
components, starting on box planes, starting far outside the scene, grazing. For both ways to build the wide tree, per-lane and packet kernels:
never a miss where the oracle hits, never a farther hit, a closer hit only by rounding (<= 1e-6 relative), (b, c, t) bit-equal on the oracle's own triangle; a hit the
reference lacks altogether only on the rays built to lie in box planes, and then a true hit of its triangle (float64 check).
Exits 1 on the first violation, printing the case."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
rt = importlib.import_module("raytracing-course-hw-public_amd")
import oracle  # noqa: E402  (checker only)


def make_scene(sg, rng, n, scale, offset, shape):
    c = rng.uniform(-1.0, 1.0, size=(n, 1, 3)) * shape
    pos = c + rng.uniform(-1.0, 1.0, size=(n, 3, 3)) * float(rng.choice([0.01, 0.1, 0.5])) * shape
    if rng.integers(0, 3) == 0:  # a share of axis-aligned (flat-boxed) triangles
        k = rng.integers(0, n, size=max(1, n // 4))
        ax = rng.integers(0, 3, size=len(k))
        pos[k, :, ax] = pos[k, 0:1, ax]
    kind = int(rng.integers(0, 8))  # now and then a pathological arrangement
    if kind == 0 and n >= 9:  # duplicates: a few distinct triangles, each many times (boxes that cannot be told apart)
        pos = pos[rng.integers(0, max(1, n // 50), size=n)]
    elif kind == 1 and n >= 9:  # one giant triangle over everything else
        pos[0] = np.array([[-3.0, -3.0, 0.0], [3.0, -3.0, 0.1], [0.0, 3.0, -0.1]]) * shape
    elif kind == 2:  # a fan: every triangle shares one vertex
        pos[:, 0, :] = pos[0, 0, :]
    elif kind == 3 and n >= 9:  # coplanar tiles on an axis-aligned grid: flat boxes, shared edges, exact ties everywhere
        g = int(np.ceil(np.sqrt(n / 2)))
        ij = np.stack(np.meshgrid(np.arange(g), np.arange(g), indexing="ij"), axis=-1).reshape(-1, 2)[: (n + 1) // 2]
        a = np.concatenate([ij, np.zeros((len(ij), 1))], axis=1) / g * 2 - np.array([1.0, 1.0, 0.0])
        q = np.stack([a, a + [2.0 / g, 0, 0], a + [0, 2.0 / g, 0]], axis=1)
        r = np.stack([a + [2.0 / g, 2.0 / g, 0], a + [0, 2.0 / g, 0], a + [2.0 / g, 0, 0]], axis=1)
        pos = np.concatenate([q, r], axis=0)[:n] * shape
    elif kind == 4 and n >= 9:  # sizes growing geometrically along a line
        s_ = np.exp(np.linspace(np.log(1e-4), 0.0, n))[:, None, None]
        pos = (np.cumsum(s_, axis=0) / s_.sum() * 2 - 1) * np.array([1.0, 0.0, 0.0]) + rng.uniform(-1.0, 1.0, size=(n, 3, 3)) * s_
        pos = pos * shape
    pos = (pos * scale + offset).astype(np.float32)
    tang = np.tile(np.array([1, 0, 0], dtype=np.float32), (n, 3, 1))
    mats = [sg.Material(color=(0.7, 0.6, 0.5, 1.0), roughness=0.6, metallic=0.2)]
    cam = sg.look_camera(tuple(float(v) for v in (np.asarray(offset) + np.array([0.0, 0.0, 3.0]) * scale)), yaw_deg=0.0, yfov=0.9)
    return sg.Scene(positions=pos, normals=None, texcoords=np.zeros((n, 3, 2), np.float32), tangents=tang, material_ids=np.zeros(n, np.uint32), materials=mats, textures=[], camera=cam)


def make_rays(rng, sc, n, scale, offset, shape):
    lo, hi = sc.positions.reshape(-1, 3).min(axis=0).astype(np.float64), sc.positions.reshape(-1, 3).max(axis=0).astype(np.float64)
    ext = np.maximum(hi - lo, 1e-30)
    o = rng.uniform(lo - 0.5 * ext, hi + 0.5 * ext, size=(n, 3))
    tgt = sc.positions.reshape(-1, 3)[rng.integers(0, sc.positions.shape[0] * 3, size=n)].astype(np.float64) + rng.normal(size=(n, 3)) * 0.02 * ext
    d = tgt - o
    kind = rng.integers(0, 10, size=n)
    ax = rng.integers(0, 3, size=n)
    idx = np.arange(n)
    par = kind == 0  # axis-parallel
    d[par] = 0.0
    d[par, ax[par]] = rng.choice([-1.0, 1.0], size=int(par.sum()))
    o[par] = tgt[par]
    o[par, ax[par]] = np.where(d[par, ax[par]] > 0, lo[ax[par]] - ext[ax[par]], hi[ax[par]] + ext[ax[par]])
    z = kind == 1  # one component exactly zero
    d[z, ax[z]] = 0.0
    tiny = kind == 2  # one component denormal-small relative to the others
    d[tiny, ax[tiny]] *= 1e-30
    onp = kind == 3  # origin exactly on a scene-box plane
    o[onp, ax[onp]] = np.where(rng.integers(0, 2, size=int(onp.sum())) == 0, lo[ax[onp]], hi[ax[onp]])
    far = kind == 4  # origin far outside
    o[far] = tgt[far] - d[far] / np.maximum(np.linalg.norm(d[far], axis=1, keepdims=True), 1e-300) * ext.max() * float(rng.choice([3.0, 10.0, 30.0]))  # (from 10^5 extents away float32 cannot tell a hit from a miss: the reference's own outcome is noise there)
    d[far] = tgt[far] - o[far]
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    nrm[nrm == 0] = 1.0
    d = d / nrm
    bad = ~np.isfinite(d).all(axis=1) | (np.abs(d).sum(axis=1) == 0)
    d[bad] = np.array([0.0, 0.0, -1.0])
    return np.concatenate([o, d], axis=1).astype(np.float32), kind <= 3  # (rays, which of them are degenerate by construction)


def true_hit(P, ray, prim, t):
    """float64 Cramer solution of `ray` against triangle `prim`: inside (1e-5 slack) and at the reported distance (1e-5 relative)."""
    a, b, c = P[prim]
    oo, dd = ray[:3].astype(np.float64), ray[3:].astype(np.float64)
    edge = max(np.linalg.norm(b - a), np.linalg.norm(c - a), np.linalg.norm(c - b))
    mag = max(np.abs(P[prim]).max(), np.abs(oo).max())
    if edge == 0.0 or np.linalg.norm(np.cross(b - a, c - a)) / edge < 64.0 * float(np.spacing(np.float32(mag))):
        return True  # a sliver thinner than the coordinates resolve: inside / outside is not decidable from the float32 vertices
    m = np.stack([b - a, c - a, -dd], axis=1)
    try:
        x = np.linalg.solve(m, oo - a)
    except np.linalg.LinAlgError:
        return True  # the ray lies in the triangle's plane: nothing to decide in float64 either
    return x[0] >= -1e-5 and x[1] >= -1e-5 and x[0] + x[1] <= 1 + 1e-5 and abs(x[2] - t) <= 1e-5 * abs(t)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    sg = rt.scenegen
    t0 = time.time()
    refused = closer_total = ties_total = new_total = grazing = 0
    for case in range(n_cases):
        n = int(rng.choice([1, 3, 9, 40, 300, 3000, 20000, 20000, 250000]))
        k = int(rng.integers(-30, 41))
        scale = float(2.0 ** k)
        shape = np.array(rng.choice([[1, 1, 1], [1, 1, 1e-3], [1, 1e-4, 1e-4], [1, 0.3, 0.05]]), dtype=np.float64)
        off_mag = float(rng.choice([0.0, 1.0, 64.0, 4096.0]))
        offset = rng.uniform(-1, 1, size=3) * off_mag * scale
        print(f"case {case:3d}: n {n:5d} scale 2^{k:<3d} shape {shape.tolist()} offset x{off_mag:g} ...", end=" ", flush=True)
        sc = make_scene(sg, rng, n, scale, offset, shape)
        rays, degenerate = make_rays(rng, sc, 6000, scale, offset, shape)
        P = sc.positions.astype(np.float64)
        orc = oracle.OracleScene(sc)
        try:
            op, ob = orc.cast_rays(rays)
            msgs = []
            for kw in (dict(wide=True), dict(wide=True, device_bvh=True)):
                try:
                    dev = rt.DeviceScene(sc, **kw)
                except rt.RtError as e:  # the documented refusals (exponent range) are fine; anything else is not
                    if "exponent range" in str(e):
                        refused += 1
                        msgs.append("refused (exponent range)")
                        continue
                    raise
                try:
                    for mode in (rt.RT_CAST_EXTEND, rt.RT_CAST_PACKET):
                        gp, gb, _ = dev.cast_rays_ex(rays, mode)
                        miss_o, miss_g = op == 0xFFFFFFFF, gp == 0xFFFFFFFF
                        both = ~miss_o & ~miss_g
                        lost = int((miss_g & ~miss_o).sum())
                        farther = int((both & (gb[:, 2] > ob[:, 2])).sum())
                        ordinary = both & ~degenerate  # (on the in-plane rays the reference may have skipped a box altogether: any closer TRUE hit is legal there)
                        # "closer by rounding": 1e-6 relative, plus what the coordinates' own resolution allows — a scene 4096 extents from the origin has
                        # coordinates whose ulp is 1e-4 of its extent, and the reference's box-entry and triangle distances then disagree by that much
                        mag = np.maximum(np.abs(rays[:, :3]).max(axis=1), float(np.abs(sc.positions).max())).astype(np.float64)
                        allow = 1e-6 + 8.0 * 2.0 ** -24 * mag[ordinary] / np.maximum(ob[ordinary, 2].astype(np.float64), 1e-300)
                        rel = (ob[ordinary, 2].astype(np.float64) - gb[ordinary, 2].astype(np.float64)) / np.maximum(ob[ordinary, 2].astype(np.float64), 1e-300) - allow + 1e-6
                        closer = int((both & (gb[:, 2] < ob[:, 2])).sum()) + int((miss_o & ~miss_g).sum())
                        same = both & (op == gp)
                        bits_ok = np.array_equal(gb[same].view(np.uint32), ob[same].view(np.uint32))
                        ties = int((both & (gb[:, 2] == ob[:, 2]) & (op != gp)).sum())
                        closer_total += closer
                        ties_total += ties
                        # a hit the reference does not have at all is legal only on the rays built to lie IN box planes (zero direction components, origins on
                        # planes: the reference's outcome there hangs on 0/0 = NaN compare order, bvh.h:141-145), and it must be a true hit of its triangle
                        new_hit = miss_o & ~miss_g
                        fake = [int(i) for i in np.flatnonzero(new_hit | (both & (gb[:, 2] < ob[:, 2])))[:300] if not true_hit(P, rays[i], int(gp[i]), float(gb[i, 2]))]
                        new_total += int(new_hit.sum())
                        if lost or farther or rel.max(initial=0.0) > 1e-6 or not bits_ok or int((new_hit & ~degenerate).sum()) or fake:
                            print(f"case {case}: n {n} scale 2^{k} shape {shape.tolist()} offset {offset.tolist()} {kw} mode {mode}: lost {lost} farther {farther} max closer rel {rel.max(initial=0.0):.3e} "
                                  f"bits_ok {bits_ok} new hits {int(new_hit.sum())} (on ordinary rays {int((new_hit & ~degenerate).sum())}) not true hits {fake[:5]}", flush=True)
                            sel = np.flatnonzero((miss_g & ~miss_o) | (both & (gb[:, 2] > ob[:, 2])) | (new_hit & ~degenerate))
                            i = int(sel[0]) if len(sel) else (fake[0] if fake else -1)
                            if i >= 0:
                                print(f"   ray {i}: {rays[i].tolist()} oracle prim {op[i]} bct {ob[i].tolist()} gpu prim {gp[i]} bct {gb[i].tolist()}")
                            sys.exit(1)
                    msgs.append("ok")
                finally:
                    dev.close()
            # the binary device trees (PLOC, radix) under the reference's traversal: hit / miss as the oracle, t as the oracle's up to the coordinates' resolution
            # (which of several candidates within rounding the reference's pruning rule finds depends on the tree: DESIGN.md, production contract)
            for kw in (dict(device_bvh=True), dict(device_bvh=True, device_builder=rt.RT_BUILDER_LBVH)):
                dev = rt.DeviceScene(sc, **kw)
                try:
                    for mode in (rt.RT_CAST_EXTEND, rt.RT_CAST_EXTEND_GLOBAL):
                        gp, gb, _ = dev.cast_rays_ex(rays, mode)
                        miss_o, miss_g = op == 0xFFFFFFFF, gp == 0xFFFFFFFF
                        both = ~miss_o & ~miss_g & ~degenerate
                        mag = np.maximum(np.abs(rays[:, :3]).max(axis=1), float(np.abs(sc.positions).max())).astype(np.float64)
                        tol = 4.0 * np.spacing(ob[both, 2]).astype(np.float64) + 8.0 * 2.0 ** -24 * mag[both]
                        dt = np.abs(gb[both, 2].astype(np.float64) - ob[both, 2].astype(np.float64))
                        # a ray that grazes a triangle's EDGE within rounding (a barycentric coordinate of either answer within 1e-5 of zero) hits that triangle or not
                        # depending on whether the triangle's leaf box lets the ray in at all, i.e. on the tree: either answer is the reference arithmetic's
                        edge_o = np.minimum(np.minimum(np.abs(ob[both, 0]), np.abs(ob[both, 1])), np.abs(1.0 - ob[both, 0] - ob[both, 1])) < 1e-5
                        edge_g = np.minimum(np.minimum(np.abs(gb[both, 0]), np.abs(gb[both, 1])), np.abs(1.0 - gb[both, 0] - gb[both, 1])) < 1e-5
                        grazing += int(((dt > tol) & (edge_o | edge_g)).sum())
                        dt = np.where(edge_o | edge_g, 0.0, dt)
                        flips = int(((miss_o != miss_g) & ~degenerate).sum())
                        if flips or (dt > tol).any():
                            i = int(np.flatnonzero((miss_o != miss_g) & ~degenerate)[0]) if flips else int(np.flatnonzero(both)[np.argmax(dt - tol)])
                            print(f"\ncase {case}: binary tree {kw} mode {mode}: hit/miss flips on ordinary rays {flips}, worst |dt| - tol {float((dt - tol).max(initial=0.0)):.3e}", flush=True)
                            print(f"   ray {i}: {rays[i].tolist()} oracle prim {op[i]} bct {ob[i].tolist()} gpu prim {gp[i]} bct {gb[i].tolist()}")
                            sys.exit(1)
                    msgs.append("bin ok")
                finally:
                    dev.close()
            print(f"hits {int((op != 0xFFFFFFFF).sum())}/6000 -> {msgs}", flush=True)
        finally:
            orc.close()
    print(f"{n_cases} cases: no lost hit, no farther hit, closer hits {closer_total} (all <= 1e-6 relative, incl. {new_total} hits only the wide tree has: in-plane rays, each a true hit), exact ties {ties_total}, refused builds {refused}; binary trees: {grazing} edge-grazing rays with another answer; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

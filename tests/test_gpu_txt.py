"""Scene-txt scenes on the GPU (BASELINE configs 1-2): triangles from BOX / TRIANGLE primitives through the BVH
path, ELLIPSOID / PLANE through the brute-force analytic-primitive kernel (wf_extend_prims; include/rt_primspec.h), against
the CPU oracle on the same loaded scene. Analytic primitives are "parity unpinned" against the reference (HEAD has none);
the bar here is oracle == GPU bit for bit, in both schedules (wavefront pipeline and megakernel), AND an independent float64 solution written in
this file (hit / miss, primitive, distance within a stated bound, normals)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TXT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "txt")
COUNTERS = ("samples", "casts", "nodes_visited", "box_tests", "tri_tests", "shaded_hits", "light_queries", "light_tri_tests", "light_hits")

SPHERES_ONLY = """DIMENSIONS 512 512
RAY_DEPTH 6
SAMPLES 64
BG_COLOR 0.6 0.7 0.9
CAMERA_POSITION 0 1 12
CAMERA_RIGHT 1 0 0
CAMERA_UP 0 1 0
CAMERA_FORWARD 0 0 -1
CAMERA_FOV_X 0.927295218
NEW_PRIMITIVE
PLANE 0 1 0
POSITION 0 -2 0
COLOR 0.8 0.8 0.8
NEW_PRIMITIVE
ELLIPSOID 2 2 2
POSITION -2.5 0 0
COLOR 0.9 0.3 0.3
NEW_PRIMITIVE
ELLIPSOID 1.5 2.5 1
POSITION 2 0.5 -1
ROTATION 0.1 0.2 0.3 0.9273618
COLOR 0.3 0.9 0.4
METALLIC
NEW_PRIMITIVE
ELLIPSOID 1 1 1
POSITION 0.3 -1 3
COLOR 0.9 0.9 0.9
DIELECTRIC
IOR 1.5
NEW_PRIMITIVE
ELLIPSOID 0.7 0.7 0.7
POSITION 0 4.5 1
EMISSION 6 5 4
"""


def _pair(gpu, oracle, path):
    ls = gpu.parse_scene_txt(path)
    return gpu.DeviceScene(ls), oracle.OracleScene(ls), ls


def _rays(seed, n, lo=-6.0, hi=6.0):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


@pytest.mark.parametrize("name", ["cornell_mixed", "boxes_only"])
def test_txt_scene_matches_oracle(gpu, oracle, name):
    dev, orc, ls = _pair(gpu, oracle, os.path.join(TXT, name + ".txt"))
    try:
        W, H, SPP = 64, 56, 6
        ofb, ost = orc.run_raytracer(W, H, SPP, seed=31)
        for kw in ({}, {"megakernel": True}):
            gfb, gst = dev.run_raytracer(W, H, SPP, seed=31, counters=True, **kw)
            assert np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32)), (name, kw, int((gfb != ofb).any(axis=2).sum()))
            for k in COUNTERS:
                assert gst[k] == ost[k], (k, kw)
        rays = _rays(4, 20000)
        gp, gb = dev.cast_rays(rays)
        op, ob = orc.cast_rays(rays)
        assert np.array_equal(gp, op) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
        n_tri = ls.arrays()["positions"].shape[0]
        if name == "cornell_mixed":
            assert (gp >= n_tri).sum() > 5000 and (gp < n_tri).sum() > 500  # analytic primitives AND triangles win rays
        assert np.array_equal(dev.light_pdf(rays).view(np.uint32), orc.light_pdf(rays).view(np.uint32))
        img, _ = dev.run_raytracer_rgb8(W, H, SPP, seed=31)
        assert np.array_equal(img, oracle.tonemap(ofb))
        sh = np.zeros_like(ofb)
        for r in range(3):
            dev.run_raytracer(W, H, SPP, seed=31, shard_index=r, shard_count=3, shard_block=256, out=sh)
        assert np.array_equal(sh.view(np.uint32), ofb.view(np.uint32))
        rfb, _ = dev.run_raytracer(W, H, 2, rng_mode=gpu.RT_RNG_REFERENCE)
        orf, _ = orc.run_raytracer(W, H, 2, rng_mode=gpu.RT_RNG_REFERENCE)
        assert np.array_equal(rfb.view(np.uint32), orf.view(np.uint32))
        if name == "boxes_only":  # ... and the reference binary's own bytes for this scene (tests/test_scene_txt.py pins the fixture)
            img, _ = dev.run_raytracer_rgb8(64, 48, 4, rng_mode=gpu.RT_RNG_REFERENCE)
            gold = oracle.read_ppm(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "txt_boxes_64x48x4.ppm"))
            assert np.array_equal(img, gold)
    finally:
        dev.close()
        orc.close()


def test_config2_primitives_only_512x512(gpu, oracle, tmp_path):
    """BASELINE config 2 shape: spheres / planes only (no triangle, no BVH: the intersect kernel alone), 512x512. Bit-exact
    against the oracle at 4 SPP; at the configuration's 64 SPP the two GPU schedules must agree with each other."""
    f = tmp_path / "spheres.txt"
    f.write_text(SPHERES_ONLY)
    dev, orc, ls = _pair(gpu, oracle, str(f))
    try:
        a = ls.arrays()
        assert a["positions"].shape[0] == 0 and len(a["primitives"]) == 5
        W = H = 512
        ofb, ost = orc.run_raytracer(W, H, 4, seed=5)
        gfb, gst = dev.run_raytracer(W, H, 4, seed=5, counters=True)
        assert np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32))
        assert gst["casts"] == ost["casts"] and gst["shaded_hits"] == ost["shaded_hits"] and gst["nodes_visited"] == 0
        full, fst = dev.run_raytracer(W, H, 64, seed=5)
        mega, _ = dev.run_raytracer(W, H, 64, seed=5, megakernel=True)
        assert np.array_equal(full.view(np.uint32), mega.view(np.uint32)) and np.isfinite(full).all()
        assert fst["samples"] == W * H * 64
        gp, gb = dev.cast_rays(_rays(9, 5000, -8, 8))
        op, ob = orc.cast_rays(_rays(9, 5000, -8, 8))
        assert np.array_equal(gp, op) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    finally:
        dev.close()
        orc.close()


def test_config2_practice3_5_through_the_hip_path(gpu, oracle, tmp_path):
    """BASELINE config 2 on one of the files it names: sample_data/homebrew_primitives/practice3_5.txt (committed as tests/golden/txt/practice3_5.txt:
    a Cornell box of 5 PLANEs, an emissive BOX, a rotated BOX and an ELLIPSOID), at the file's own 512x512, 64 SPP, ray depth 6, through librt_amd.so:
    (a) the CLI writes the oracle's PPM; (b) the device-RNG framebuffer (16.8 M samples) and every event counter equal the oracle's, in both schedules,
    and the device film's bytes equal the host film of the oracle's image; (c) reference-RNG mode equals the oracle at 8 SPP; (d) hit records on 20 000 rays."""
    from conftest import PRACTICE3_5

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev, orc, ls = _pair(gpu, oracle, PRACTICE3_5)
    try:
        info, a = ls.info(), ls.arrays()
        W, H, SPP = info["width"], info["height"], info["samples"]
        assert (W, H, SPP, int(a["ray_depth"])) == (512, 512, 64, 6) and a["positions"].shape[0] == 24 and [p["kind"] for p in a["primitives"]] == [2, 2, 2, 2, 2, 1]
        ofb, ost = orc.run_raytracer(W, H, SPP, seed=23)
        out = tmp_path / "c2.ppm"
        subprocess.check_call([os.path.join(root, "run.sh"), PRACTICE3_5, str(W), str(H), str(SPP), str(out)], env=dict(os.environ, RT_RNG_MODE="device", RT_SEED="23", RT_DEVICE="0"))
        assert np.array_equal(oracle.read_ppm(str(out)), oracle.tonemap(ofb))
        for kw in ({}, {"megakernel": True}):
            gfb, gst = dev.run_raytracer(W, H, SPP, seed=23, counters=True, **kw)
            assert np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32)), (kw, int((gfb != ofb).any(axis=2).sum()))
            for k in COUNTERS:
                assert gst[k] == ost[k], (k, kw)
        assert ost["samples"] == W * H * SPP and ost["casts"] > 3 * ost["samples"]  # a closed box: paths live for several bounces
        img, _ = dev.run_raytracer_rgb8(W, H, SPP, seed=23)
        assert np.array_equal(img, oracle.tonemap(ofb))
        rfb, _ = dev.run_raytracer(W, H, 8, rng_mode=gpu.RT_RNG_REFERENCE)
        orf, _ = orc.run_raytracer(W, H, 8, rng_mode=gpu.RT_RNG_REFERENCE)
        assert np.array_equal(rfb.view(np.uint32), orf.view(np.uint32))
        rays = _rays(8, 20000, -4.5, 4.5)
        gp, gb = dev.cast_rays(rays)
        op, ob = orc.cast_rays(rays)
        assert np.array_equal(gp, op) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
        assert (gp != 0xFFFFFFFF).mean() > 0.7 and (gp < 24).sum() > 500 and (gp == 24 + 5).sum() > 200  # (the room is open towards the camera) boxes and the ellipsoid win rays
    finally:
        dev.close()
        orc.close()


def test_config1_scene000_through_the_hip_path(gpu, sg, oracle, tmp_path):
    """BASELINE config 1: sample_data/scene-000.txt (committed as tests/golden/txt/scene-000.txt: ELLIPSOID + PLANE + BOX), 256x256, 4 SPP,
    through librt_amd.so. (a) the CLI, `run.sh scene-000.txt 256 256 4 out.ppm`, writes the oracle's PPM in both RNG modes; (b) device-RNG
    framebuffer and event counters equal the oracle's, in both schedules, sharded and through the device film; (c) the part of the scene
    the reference at HEAD can still render, the BOX's triangles exported as glTF, renders in reference-RNG mode to the bytes the UNMODIFIED
    reference binary produced (tests/golden/txt_scene000_box_64x48x4.ppm, tests/golden/make_scene000_golden.py)."""
    from conftest import SCENE000, scene000_box_gltf

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    W = H = 256
    SPP = 4
    dev, orc, ls = _pair(gpu, oracle, SCENE000)
    try:
        a = ls.arrays()
        assert a["positions"].shape[0] == 12 and [p["kind"] for p in a["primitives"]] == [1, 2]
        # (a) the CLI
        for mode, seed in (("reference", 0), ("device", 17)):
            out = tmp_path / f"c1_{mode}.ppm"
            subprocess.check_call([os.path.join(root, "run.sh"), SCENE000, str(W), str(H), str(SPP), str(out)], env=dict(os.environ, RT_RNG_MODE=mode, RT_SEED=str(seed), RT_DEVICE="0"))
            ofb, _ = orc.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_REFERENCE if mode == "reference" else gpu.RT_RNG_DEVICE, seed=seed)
            assert np.array_equal(oracle.read_ppm(str(out)), oracle.tonemap(ofb)), mode
        # (b) the library
        ofb, ost = orc.run_raytracer(W, H, SPP, seed=17)
        for kw in ({}, {"megakernel": True}, {"packet_mode": gpu.RT_PACKET_ON}):
            gfb, gst = dev.run_raytracer(W, H, SPP, seed=17, counters=True, **kw)
            assert np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32)), (kw, int((gfb != ofb).any(axis=2).sum()))
            for k in COUNTERS:
                assert gst[k] == ost[k], (k, kw)
        img, _ = dev.run_raytracer_rgb8(W, H, SPP, seed=17)
        assert np.array_equal(img, oracle.tonemap(ofb))
        sh = np.zeros_like(ofb)
        for r in range(4):
            dev.run_raytracer(W, H, SPP, seed=17, shard_index=r, shard_count=4, shard_block=8 * W, out=sh)
        assert np.array_equal(sh.view(np.uint32), ofb.view(np.uint32))
        rfb, _ = dev.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_REFERENCE)
        orf, _ = orc.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_REFERENCE)
        assert np.array_equal(rfb.view(np.uint32), orf.view(np.uint32))
        rays = _rays(6, 20000, -6, 6)
        gp, gb = dev.cast_rays(rays)
        op, ob = orc.cast_rays(rays)
        assert np.array_equal(gp, op) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
        assert (gp == 12).sum() > 500 and (gp == 13).sum() > 500 and (gp < 12).sum() > 20  # all three primitive kinds win rays
    finally:
        dev.close()
        orc.close()
    # (c) the BOX against the reference binary's bytes
    gltf, _ = scene000_box_gltf(gpu, sg, tmp_path)
    box = gpu.DeviceScene(gpu.parse_gltf_scene(gltf, 64 / 48))
    try:
        img, _ = box.run_raytracer_rgb8(64, 48, 4, rng_mode=gpu.RT_RNG_REFERENCE)
        gold = oracle.read_ppm(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "txt_scene000_box_64x48x4.ppm"))
        assert np.array_equal(img, gold)
    finally:
        box.close()


# ------------------------------------------------------------------------------------------------ independent pin of ELLIPSOID / PLANE
# The reference at HEAD has no such primitive, and the oracle compiles the same include/rt_primspec.h as the kernels: oracle == GPU
# proves the two compilers agree, not that the header is right. What follows is written from the geometry, in float64, without that header.
EPS = 1e-4


def quat_matrix(q):
    """Rotation matrix of the unit quaternion q = (x, y, z, w) — the textbook matrix form (not the cross-product form of rt_primspec.h)."""
    x, y, z, w = (float(v) for v in q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)

def closed_form(prims, rays):
    """float64 closed forms, written from the geometry (world-space quadric x^T A x = 1; point-normal plane), independent of include/rt_primspec.h.
    Returns per ray: index of the nearest primitive (-1 = miss), t, unit normal facing the ray, and a 'margin' in [0, 1] that is small when
    the answer is ill-conditioned (grazing root, root next to EPS, two primitives at almost the same distance)."""
    o = rays[:, :3].astype(np.float64)
    d = rays[:, 3:].astype(np.float64)
    n = len(rays)
    best_t = np.full(n, np.inf)
    second_t = np.full(n, np.inf)
    best_i = np.full(n, -1)
    best_n = np.zeros((n, 3))
    margin = np.ones(n)
    for i, p in enumerate(prims):
        c = np.asarray(p["position"], dtype=np.float64)
        if p["kind"] == 1:
            R = quat_matrix(np.asarray(p["rotation"], dtype=np.float64) / np.linalg.norm(np.asarray(p["rotation"], dtype=np.float64)))
            A = R @ np.diag(1.0 / np.asarray(p["param"], dtype=np.float64) ** 2) @ R.T
            x0 = o - c
            qa = np.einsum("ij,jk,ik->i", d, A, d)
            qb = 2 * np.einsum("ij,jk,ik->i", d, A, x0)
            qc = np.einsum("ij,jk,ik->i", x0, A, x0) - 1.0
            disc = qb * qb - 4 * qa * qc
            ok = disc >= 0
            sq = np.sqrt(np.where(ok, disc, 0.0))
            qq = -0.5 * (qb + np.where(qb >= 0, 1.0, -1.0) * sq)  # the cancellation-free form
            with np.errstate(divide="ignore", invalid="ignore"):
                ra, rb = qq / qa, qc / qq
            t1, t2 = np.minimum(ra, rb), np.maximum(ra, rb)
            t = np.where(t1 >= EPS, t1, np.where(t2 >= EPS, t2, np.inf))
            t = np.where(ok, t, np.inf)
            x = x0 + d * np.where(np.isfinite(t), t, 0.0)[:, None]
            g = x @ A.T
            nn = g / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-300)
            m = np.minimum(np.abs(disc) / np.maximum(qb * qb + np.abs(4 * qa * qc), 1e-300), 1.0)  # grazing
            m = np.minimum(m, np.minimum(np.abs(t1 - EPS), np.abs(t2 - EPS)) / EPS)            # a root next to the EPS threshold
        else:
            nrm = np.asarray(p["param"], dtype=np.float64)
            nrm = nrm / np.linalg.norm(nrm)
            dn = d @ nrm
            with np.errstate(divide="ignore", invalid="ignore"):
                t = ((c - o) @ nrm) / dn
            m = np.minimum(np.abs(dn), 1.0)
            m = np.minimum(m, np.abs(t - EPS) / EPS)
            t = np.where(np.isfinite(t) & (t >= EPS), t, np.inf)
            nn = np.broadcast_to(nrm, (n, 3))
        nn = np.where((np.einsum("ij,ij->i", nn, d) > 0)[:, None], -nn, nn)
        closer = t < best_t
        second_t = np.where(closer, best_t, np.minimum(second_t, t))
        best_n = np.where(closer[:, None], nn, best_n)
        best_i = np.where(closer, i, best_i)
        best_t = np.where(closer, t, best_t)
        margin = np.minimum(margin, m)
    with np.errstate(invalid="ignore"):
        sep = np.where(np.isfinite(second_t), (second_t - best_t) / np.maximum(best_t, 1e-300), 1.0)
    margin = np.minimum(margin, np.minimum(sep, 1.0))
    return best_i, best_t, best_n, margin

def length_scale(prims, rays, idx):
    """|o - c| + largest radius of the primitive that was hit: the magnitudes the float32 quadratic works with (its roots are differences of
    quantities of this size, so its absolute error scales with it, not with t)."""
    o = rays[:, :3].astype(np.float64)
    L = np.zeros(len(rays))
    for i, p in enumerate(prims):
        sel = idx == i
        r = float(np.max(np.abs(p["param"]))) if p["kind"] == 1 else 0.0
        L[sel] = np.linalg.norm(o[sel] - np.asarray(p["position"], dtype=np.float64), axis=1) + r
    return L


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_analytic_primitives_against_an_independent_float64_solution(gpu, sg, seed):
    """Five rotated / translated ellipsoids and two planes (one with a non-unit normal), 200 000 random rays through rt_cast_rays and
    rt_surface_normals against the float64 closed forms above. Away from ill-conditioned answers (margin > 1e-3: no grazing root, no root next to
    EPS, no second primitive within 0.1 % of the distance) the device must report the same hit / miss and the same primitive on EVERY ray;
    t within 1500 ulps of the length scale max(|o - c| + r_max, t) (the float32 solve restates the reference's half-b quadratic of
    raytracer.h:61-77, whose roots are differences of quantities of that size; measured: <= 310, p99.99 125); the normal within 2e-5 of the
    float64 normal AT THE DEVICE'S hit point (the normal arithmetic on its own: the hit point in the primitive's frame carries an error of an ulp
    of |o - c| ~ 10, divided by the smallest radius 0.3 -> ~ 4e-6; measured on the CPU build of the same header: <= 7.2e-6) and within 2e-4 of the
    normal at the exact root (measured 3.4e-5); always unit length and facing the ray."""
    rng = np.random.default_rng(seed)

    def rq():
        q = rng.normal(size=4)
        return tuple(float(x) for x in (q / np.linalg.norm(q)).astype(np.float32))

    prims = [dict(kind=1, material_id=0, param=tuple(float(x) for x in rng.uniform(0.3, 2.5, 3).astype(np.float32)),
                  position=tuple(float(x) for x in rng.uniform(-4, 4, 3).astype(np.float32)), rotation=rq()) for _ in range(5)]
    prims.append(dict(kind=2, material_id=0, param=(0.2, 1.0, -0.3), position=(0.0, -5.0, 0.0), rotation=(0.0, 0.0, 0.0, 1.0)))
    prims.append(dict(kind=2, material_id=0, param=(-1.0, 0.1, 0.4), position=(6.0, 0.0, 1.0), rotation=rq()))  # a plane ignores its rotation
    z = np.zeros((0, 3, 3), dtype=np.float32)
    sc = sg.Scene(positions=z, normals=z, texcoords=np.zeros((0, 3, 2), dtype=np.float32), tangents=z, material_ids=np.zeros(0, dtype=np.uint32),
                  materials=[sg.Material(color=(0.8, 0.8, 0.8, 1.0), emission=(0.0, 0.0, 0.0))], camera=sg.look_camera((0.0, 0.0, 5.0), yaw_deg=0.0, yfov=0.8), primitives=prims)
    n = 200_000
    o = rng.uniform(-8, 8, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    dev = gpu.DeviceScene(sc)
    try:
        prim, bct = dev.cast_rays(rays)
        prim2, t2, nn, sn = dev.surface_normals(rays)
        prim3, bct3, _ = dev.cast_rays_ex(rays, gpu.RT_CAST_EXTEND)  # ... and through the renderer's own kernels (wf_extend + wf_extend_prims)
    finally:
        dev.close()
    assert np.array_equal(prim, prim2) and np.array_equal(bct[:, 2].view(np.uint32), t2.view(np.uint32)) and np.array_equal(nn.view(np.uint32), sn.view(np.uint32))
    assert np.array_equal(prim, prim3) and np.array_equal(bct.view(np.uint32), bct3.view(np.uint32))
    p32 = [dict(p, param=np.float32(p["param"]), position=np.float32(p["position"]), rotation=np.float32(p["rotation"])) for p in prims]  # what the device was given
    bi, bt, bn, margin = closed_form(p32, rays)
    hit = prim != 0xFFFFFFFF
    good = margin > 1e-3
    assert good.mean() > 0.97
    assert not ((hit != (bi >= 0)) & good).any(), int(((hit != (bi >= 0)) & good).sum())
    idx = hit & (bi >= 0) & good
    assert np.array_equal(prim[idx].astype(np.int64), bi[idx])  # n_triangles == 0: primitive i is reported as i
    for k in range(len(prims)):
        assert (bi[idx] == k).sum() > 200, k  # every primitive is hit, ellipsoids from outside and inside
    t = bct[:, 2].astype(np.float64)
    L = np.maximum(length_scale(p32, rays, bi), bt)
    ulps = np.abs(t - bt)[idx] / (L[idx] * 2.0 ** -23)
    assert ulps.max() <= 1500, ulps.max()
    # normals: unit length, facing the ray, equal to the float64 normal
    r64 = rays.astype(np.float64)
    nd = nn.astype(np.float64)
    assert np.abs(np.linalg.norm(nd[hit], axis=1) - 1).max() < 1e-6
    assert (np.einsum("ij,ij->i", nd[idx], r64[idx, 3:]) <= 1e-6).all()
    assert np.abs(nd[idx] - bn[idx]).max() <= 2e-4, np.abs(nd[idx] - bn[idx]).max()
    x_dev = r64[:, :3] + r64[:, 3:] * t[:, None]
    want = np.zeros_like(nd)
    for k, p in enumerate(p32):
        sel = idx & (bi == k)
        if p["kind"] == 1:
            R = quat_matrix(p["rotation"].astype(np.float64) / np.linalg.norm(p["rotation"].astype(np.float64)))
            A = R @ np.diag(1.0 / p["param"].astype(np.float64) ** 2) @ R.T
            g = (x_dev[sel] - p["position"].astype(np.float64)) @ A.T
            g /= np.linalg.norm(g, axis=1, keepdims=True)
        else:
            g = np.broadcast_to(p["param"].astype(np.float64) / np.linalg.norm(p["param"].astype(np.float64)), (int(sel.sum()), 3)).copy()
        flip = np.einsum("ij,ij->i", g, r64[sel, 3:]) > 0
        g[flip] = -g[flip]
        want[sel] = g
    worst = np.abs(nd[idx] - want[idx]).max()
    print(f"seed {seed}: t error max {ulps.max():.0f} ulps of the length scale (p99.99 {np.quantile(ulps, 0.9999):.0f}); normal error at the device's hit point {worst:.2e}, at the exact root {np.abs(nd[idx] - bn[idx]).max():.2e}")
    assert worst <= 2e-5, worst


def test_cli_renders_a_scene_txt(gpu, oracle, tmp_path):
    """run.sh <scene.txt> <W> <H> <SPP> <out.ppm>: the reference's CLI shape with the scene-txt front end behind it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(TXT, "cornell_mixed.txt")
    out = tmp_path / "c.ppm"
    env = dict(os.environ, RT_SEED="9", RT_RNG_MODE="device", RT_DEVICE="0")
    subprocess.check_call([os.path.join(root, "run.sh"), src, "48", "40", "3", str(out)], env=env)
    ls = gpu.parse_scene_txt(src)
    fb, _ = oracle.OracleScene(ls).run_raytracer(48, 40, 3, rng_mode=gpu.RT_RNG_DEVICE, seed=9)
    assert np.array_equal(oracle.read_ppm(str(out)), oracle.tonemap(fb))
    bad = tmp_path / "bad.txt"
    bad.write_text("NEW_PRIMITIVE\nTORUS 1 2\n")
    r = subprocess.run([os.path.join(root, "run.sh"), str(bad), "8", "8", "1", str(out)], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "unknown command 'TORUS'" in r.stderr


@pytest.mark.parametrize("k", [0, 1, 2])
def test_device_ellipsoid_solve_is_the_reference_sphere_routine(gpu, sg, k):
    """The device's analytic-primitive kernel against the reference's own intersect_ray_sphere (raytracer.h:61-77) for spheres at the origin:
    the same known answers the oracle is held to in tests/test_scene_txt.py, without the oracle in between."""
    from test_scene_txt import sphere_kat_check

    kat = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sphere_kat.npz"))

    def cast(scene, rays):
        dev = gpu.DeviceScene(scene)
        try:
            return dev.cast_rays(rays)
        finally:
            dev.close()

    sphere_kat_check(cast, sg, kat, k)

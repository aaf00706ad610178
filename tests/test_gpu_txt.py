"""Scene-txt scenes on the GPU (BASELINE configs 1-2): triangles from BOX / TRIANGLE primitives through the BVH
path, ELLIPSOID / PLANE through the brute-force analytic-primitive kernel (wf_extend_prims; include/rt_primspec.h), against
the CPU oracle on the same loaded scene. Analytic primitives are "parity unpinned" against the reference (HEAD has none);
the bar here is oracle == GPU bit for bit, in both schedules (wavefront pipeline and megakernel)."""
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

"""GPU parity tests: the HIP path through the C-ABI against the CPU oracle on the same seeded inputs.

Bar (BASELINE north_star): bit-exact primitive-hit indices; per-pixel radiance within 1e-5 relative. In practice the
kernels evaluate the same IEEE operation sequence as the oracle, so most comparisons below demand bit equality and
only the framebuffer test states the 1e-5 tolerance.
"""
import os

import numpy as np
import pytest

from conftest import random_rays

pytestmark = pytest.mark.gpu


# Tuning travels in rt_params (ABI 4: packet_mode, sort_mode, max_paths), not in the environment: the library reads no variable.

REL_TOL = 1e-5  # north_star: per-pixel radiance within 1e-5 relative


def _rel_err(a, b):
    den = np.maximum(np.abs(b), 1e-6)
    return np.abs(a - b) / den


@pytest.fixture(scope="module")
def pairs(gpu, oracle, scenes):
    out = {}
    for name, sc in scenes.items():
        out[name] = (gpu.DeviceScene(sc), oracle.OracleScene(sc), sc)
    yield out
    for d, o, _ in out.values():
        d.close()
        o.close()


@pytest.mark.parametrize("name", ["room_plain", "room_textured", "open_nolight", "boxes", "room_manylights"])
def test_bvh_topology_matches_oracle(pairs, name):
    dev, orc, _ = pairs[name]
    for which in (0, 1):
        a, b = dev.bvh_info(which), orc.bvh_info(which)
        assert a["root"] == b["root"]
        assert np.array_equal(a["order"], b["order"])
        assert np.array_equal(a["nodes"], b["nodes"])


@pytest.mark.parametrize("name", ["room_plain", "room_textured", "open_nolight", "boxes", "room_manylights"])
def test_closest_hit_bit_exact(pairs, name):
    dev, orc, sc = pairs[name]
    rays = random_rays(sc, 20000, seed=101)
    gp, gb = dev.cast_rays(rays)
    op, ob = orc.cast_rays(rays)
    assert np.array_equal(gp, op), f"{int((gp != op).sum())} hit-index mismatches"
    assert np.array_equal(gb.view(np.uint32), ob.view(np.uint32)), "b/c/t differ in bits"
    assert (gp != 0xFFFFFFFF).sum() > (30 if name == "open_nolight" else 1000)  # the test actually hits things


def test_closest_hit_degenerate_rays(pairs):
    """Axis-parallel rays from points ON box planes / vertices: 0/0 and +-inf slab terms (bvh.h:141-145)."""
    dev, orc, sc = pairs["boxes"]
    verts = sc.positions.reshape(-1, 3)[:300]
    rays = []
    for ax in range(3):
        for sgn in (-1.0, 1.0):
            d = np.zeros(3, dtype=np.float32)
            d[ax] = sgn
            for v in verts:
                rays.append(np.concatenate([v + np.float32(0.0), d]))
    rays = np.asarray(rays, dtype=np.float32)
    gp, gb = dev.cast_rays(rays)
    op, ob = orc.cast_rays(rays)
    assert np.array_equal(gp, op)
    assert np.array_equal(gb.view(np.uint32), ob.view(np.uint32))


def test_closest_hit_fast_division_boundaries(pairs):
    """Rays on both sides of the exact-reciprocal-division preconditions (rt_kernels.hip div_exact_fast): origin
    components that are 0, tiny (1e-30), huge (1e15); direction components that are 0, 1e-20 or dominate."""
    dev, orc, sc = pairs["room_plain"]
    rays = random_rays(sc, 6000, seed=303)
    rng = np.random.default_rng(8)
    specials_o = np.array([0.0, 1e-30, -1e-30, 1e-13, 1e15, 4.0, -20.0, 16.0], dtype=np.float32)
    specials_d = np.array([0.0, 1e-20, -1e-20, 1e-13, 1.0], dtype=np.float32)
    for i in range(3000):
        rays[i, rng.integers(0, 3)] = rng.choice(specials_o)
        if i % 2:
            rays[i, 3 + rng.integers(0, 3)] = rng.choice(specials_d)
    gp, gb = dev.cast_rays(rays)
    op, ob = orc.cast_rays(rays)
    assert np.array_equal(gp, op)
    assert np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    assert (gp != 0xFFFFFFFF).sum() > 2000


@pytest.mark.parametrize("name", ["room_plain", "room_textured", "boxes", "room_manylights"])
def test_light_pdf_bit_exact(pairs, name):
    dev, orc, sc = pairs[name]
    rays = random_rays(sc, 20000, seed=77)
    # aim half of the rays at light triangles so the sum is non-trivial
    lights = [i for i, m in enumerate(sc.material_ids) if any(e != 0 for e in sc.materials[m].emission)]
    cen = sc.positions[lights].mean(axis=1)
    tgt = cen[np.random.default_rng(5).integers(0, len(cen), size=len(rays) // 2)]
    d = tgt - rays[: len(tgt), :3]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays[: len(tgt), 3:] = d.astype(np.float32)
    g = dev.light_pdf(rays)
    o = orc.light_pdf(rays)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
    assert (o > 0).sum() > 100


@pytest.mark.parametrize("name", ["room_plain", "room_textured", "open_nolight", "boxes", "room_manylights"])
def test_render_device_rng_matches_oracle(pairs, gpu, name):
    """RT_RNG_DEVICE: the same xoshiro stream per (pixel, sample) on both sides, the reference's arithmetic otherwise (the oracle calls libm's
    sinf / cosf, the device evaluates their restatement) -> radiance within 1e-5 relative (observed: bit-identical), identical event counters."""
    dev, orc, _ = pairs[name]
    W, H, SPP = 48, 40, 6
    ofb, ost = orc.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_DEVICE, seed=1234)
    for pkt in (gpu.RT_PACKET_OFF, gpu.RT_PACKET_ON):  # primary rays per lane (wf_extend) / as packets (wf_extend_packet)
        gfb, gst = dev.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_DEVICE, seed=1234, counters=True, packet_mode=pkt)
        assert gst["packet_passes"] == (gst["passes"] if pkt == gpu.RT_PACKET_ON else 0) and gst["passes"] == 1
        assert np.isfinite(gfb).all()
        err = _rel_err(gfb, ofb)
        assert err.max() <= REL_TOL, f"max rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
        for k in ("samples", "casts", "nodes_visited", "box_tests", "tri_tests", "shaded_hits", "light_queries", "light_nodes", "light_box_tests",
                  "light_tri_tests", "light_hits", "texel_fetches"):
            assert gst[k] == ost[k], f"counter {k}: gpu {gst[k]} oracle {ost[k]} (packet {pkt})"
        assert np.array_equal(gpu.tonemap(gfb), gpu.tonemap(ofb))


@pytest.mark.parametrize("shape", [(5, 3, 1), (17, 13, 3), (64, 64, 1), (80, 60, 1), (96, 50, 2), (33, 31, 7)])
def test_queue_shapes_match_oracle(pairs, gpu, shape):
    """wf_shade appends to 64 sub-queues whose regions are whole wave slots; the next bounce maps the dense ray index back to
    a slot (with a sort from 4096 rays up, directly below). Path counts below one wave, not multiples of 64, exactly at and
    across the 4096-ray sort threshold (a queue that shrinks below it between bounces) all give the oracle's framebuffer
    bit for bit, with every ray-order key (rt_params.sort_mode, RT_SORT_*; OCTANT_CELL_CONE is what AUTO picks) and with sorting off."""
    dev, orc, _ = pairs["room_manylights"]
    W, H, SPP = shape
    ofb, ost = orc.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_DEVICE, seed=99)
    all_keys = (gpu.RT_SORT_OCTANT_FINE_CELL_CONE, gpu.RT_SORT_OCTANT_CELL_CONE, gpu.RT_SORT_CELL_OCTANT_CONE, gpu.RT_SORT_OCTANT_CELL, gpu.RT_SORT_COARSE_CELL_DIR, gpu.RT_SORT_CELL_OCTANT, gpu.RT_SORT_OFF,
                gpu.RT_SORT_AUTO)
    for sort in all_keys if shape == (96, 50, 2) else (gpu.RT_SORT_OCTANT_CELL_CONE, gpu.RT_SORT_OFF):
        gfb, gst = dev.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_DEVICE, seed=99, counters=True, sort_mode=sort)
        assert np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32)), (shape, sort)
        assert gst["samples"] == ost["samples"] == W * H * SPP and gst["casts"] == ost["casts"] and gst["shaded_hits"] == ost["shaded_hits"]


@pytest.mark.parametrize("name", ["room_textured", "room_manylights"])
def test_class_sorted_shading_windows_match_oracle(pairs, gpu, name):
    """From 2 M sorted rays up (8 blocks per CU x 1024 positions) a wave of wf_shade takes 256 queue positions at a time and hands them to its
    lanes sorted by the sampler class the ray-order sort carried along (rt_wavefront.hip, RT_SHADE_RECLASS). Which lane shades a hit must not
    change a bit: 3.07 M paths (not a multiple of 256: the last window is ragged) against the oracle, framebuffer bit for bit and every event
    counter, with the counting and the plain kernel variants; sorting off takes the 64-position path over the same queue."""
    dev, orc, _ = pairs[name]
    W, H, SPP = 641, 599, 8
    ofb, ost = orc.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_DEVICE, seed=4321)
    for sort, counters in ((gpu.RT_SORT_OCTANT_CELL_CONE, True), (gpu.RT_SORT_OCTANT_CELL_CONE, False), (gpu.RT_SORT_OFF, False)):
        gfb, gst = dev.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_DEVICE, seed=4321, counters=counters, sort_mode=sort)
        assert np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32)), (name, sort, counters)
        if counters:
            assert gst["casts"] >= 3 * W * H * SPP  # closed rooms, ~3.9 casts per sample: the first bounces hold well over 2 M rays, the windows were on
            for k in ("samples", "casts", "nodes_visited", "box_tests", "tri_tests", "shaded_hits", "light_queries", "light_nodes", "light_box_tests",
                      "light_tri_tests", "light_hits", "texel_fetches"):
                assert gst[k] == ost[k], f"counter {k}: gpu {gst[k]} oracle {ost[k]}"


@pytest.mark.parametrize("name", ["room_plain", "room_textured", "room_manylights"])
def test_render_reference_rng_matches_oracle(pairs, gpu, name):
    """RT_RNG_REFERENCE: the reference's minstd stream per 256-pixel span, one lane per span, and the reference's
    std::sin / std::cos: the oracle calls the host's glibc (as the reference binary does), the device evaluates the
    restatement of glibc's sinf / cosf (include/rt_devspec.h rt_sincos_libm). Bit-identical framebuffers."""
    dev, orc, _ = pairs[name]
    W, H, SPP = 48, 40, 3
    gfb, _ = dev.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_REFERENCE)
    ofb, _ = orc.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_REFERENCE)
    assert np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32))
    assert np.array_equal(gpu.tonemap(gfb), gpu.tonemap(ofb))


def test_wavefront_equals_megakernel_and_is_pass_invariant(pairs, gpu, monkeypatch):
    """The wavefront pipeline (default for RT_RNG_DEVICE) and the persistent megakernel (RT_FLAG_MEGAKERNEL) are two
    schedules of the same per-path arithmetic: framebuffers and event counters must be identical, and splitting the
    render into more passes (sample ranges x pixel tiles, rt_params.max_paths) must not change a bit; the progress callback
    (rt_params.progress; the reference prints one line per finished span, raytracer.h:647) reports every pass once, in order."""
    dev, orc, _ = pairs["room_textured"]
    W, H, SPP = 56, 44, 7
    ofb, _ = orc.run_raytracer(W, H, SPP, seed=21)
    mfb, mst = dev.run_raytracer(W, H, SPP, seed=21, megakernel=True, counters=True)
    wfb, wst = dev.run_raytracer(W, H, SPP, seed=21, counters=True)
    assert np.array_equal(mfb.view(np.uint32), ofb.view(np.uint32))
    assert np.array_equal(wfb.view(np.uint32), ofb.view(np.uint32))
    for k in ("samples", "casts", "nodes_visited", "box_tests", "tri_tests", "shaded_hits", "light_queries", "light_tri_tests", "texel_fetches"):
        assert mst[k] == wst[k], k
    for max_paths in (5000, 1500, 1024):  # several sample passes; < W*H -> pixel tiles as well
        seen = []
        fb, st = dev.run_raytracer(W, H, SPP, seed=21, max_paths=max_paths, progress=lambda done, total: seen.append((done, total)))
        assert np.array_equal(fb.view(np.uint32), ofb.view(np.uint32)), max_paths
        assert st["passes"] > 1 and seen == [(k + 1, st["passes"]) for k in range(st["passes"])], (max_paths, st["passes"], seen)
        sh = np.zeros_like(fb)
        for r in range(3):
            dev.run_raytracer(W, H, SPP, seed=21, shard_index=r, shard_count=3, shard_block=256, out=sh, max_paths=max_paths)
        assert np.array_equal(sh.view(np.uint32), ofb.view(np.uint32)), max_paths


def test_packet_census_reaches_the_host_and_drives_the_policy(pairs, gpu, oracle, sg):
    """wf_extend_packet counts trips and lanes served; the host reads the two words one bounce late and drops the packet kernel for a
    configuration whose packets fall apart (rt_params.packet_min_lanes). Round 3 kept the census inside the XCD ticket lines, where
    wf_advance's ticket reset wiped it before the read-back: rt_stats said 0 and the policy never engaged. Here: the census
    arrives (also with ray_depth <= 2, which has no bounce 2 to read it at), a threshold of 64 lanes switches packets off after the first
    pass, and none of it changes a bit of the image."""
    dev, orc, _ = pairs["room_textured"]
    W, H = 64, 48
    ofb, _ = orc.run_raytracer(W, H, 64, seed=3)
    fb, st = dev.run_raytracer(W, H, 64, seed=3)  # one pass of 64 SPP: RT_PACKET_AUTO starts with packets (>= 16 samples per pixel and pass)
    assert st["passes"] == 1 and st["packet_passes"] == 1 and 100 <= st["packet_lanes_x100"] <= 6400, st
    assert np.array_equal(fb.view(np.uint32), ofb.view(np.uint32))
    fb, st = dev.run_raytracer(W, H, 64, seed=3, max_paths=W * H * 16, packet_min_lanes=64.0)  # 4 passes; no packet serves 64 lanes on every trip
    assert st["passes"] == 4 and st["packet_passes"] == 1 and 100 <= st["packet_lanes_x100"] < 6400, st
    assert np.array_equal(fb.view(np.uint32), ofb.view(np.uint32))
    fb, st = dev.run_raytracer(W, H, 64, seed=3, max_paths=W * H * 16, packet_min_lanes=1.0)  # a threshold every packet meets: packets stay
    assert st["passes"] == 4 and st["packet_passes"] == 4, st
    assert np.array_equal(fb.view(np.uint32), ofb.view(np.uint32))
    fb, st = dev.run_raytracer(W, H, 64, seed=3, max_paths=W * H * 16, packet_mode=gpu.RT_PACKET_OFF)
    assert st["passes"] == 4 and st["packet_passes"] == 0 and st["packet_lanes_x100"] != 0  # (the last census read stays on record)
    for depth in (1, 2):
        sc = sg.room_scene(300, seed=5, n_lights=3, n_materials=5, tex_size=8, n_tex_sets=2, alpha_fraction=0.3)
        sc.ray_depth = depth
        d2, o2 = gpu.DeviceScene(sc), oracle.OracleScene(sc)
        try:
            want, _ = o2.run_raytracer(40, 36, 32, seed=2)
            got, st = d2.run_raytracer(40, 36, 32, seed=2, max_paths=40 * 36 * 16, packet_min_lanes=64.0)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), depth
            assert st["passes"] == 2 and st["packet_passes"] == 1 and st["packet_lanes_x100"] != 0, (depth, st)
        finally:
            d2.close()
            o2.close()


def test_full_size_bench_scene_parity(gpu, oracle, sg):
    """BASELINE config 3 at full size (S-sponza: 262 172 triangles, 16 texture sets, 1000x1000): the whole image at
    1 SPP against the oracle (bit-exact framebuffer + identical event counters), 60 000 random rays for bit-exact hit
    records, and the size-independent properties at 8 SPP: wavefront == megakernel, union of 4 shards == single
    render, ray ordering on/off and pass size do not change a bit."""
    import os

    # tex_size 1024 = exactly bench.py's scene: the 268 MB interleaved texel pool and its tiling are the tested ones
    sc = sg.room_scene(262144, seed=0x5EED5EED, tex_size=1024, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0,
                       alpha_fraction=0.02, offset=0.15, camera=sg.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
    dev = gpu.DeviceScene(sc)
    orc = oracle.OracleScene(sc)
    W = H = 1000
    gfb, gst = dev.run_raytracer(W, H, 1, seed=0x5EED5EED, counters=True)
    ofb, ost = orc.run_raytracer(W, H, 1, seed=0x5EED5EED)
    assert np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32)), f"{int((gfb != ofb).any(axis=2).sum())} of 10^6 pixels differ"
    for k in ("samples", "casts", "nodes_visited", "box_tests", "tri_tests", "shaded_hits", "light_queries", "light_tri_tests", "light_hits", "texel_fetches"):
        assert gst[k] == ost[k], k
    assert gst["casts"] > 3_000_000 and gst["nodes_visited"] > 300_000_000
    rays = random_rays(sc, 60000, seed=4242)
    gp, gb = dev.cast_rays(rays)
    op, ob = orc.cast_rays(rays)
    assert np.array_equal(gp, op) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    orc.close()
    # size-independent properties at 8 SPP (8 M paths)
    a, _ = dev.run_raytracer(W, H, 8, seed=7)
    m, _ = dev.run_raytracer(W, H, 8, seed=7, megakernel=True)
    assert np.array_equal(a.view(np.uint32), m.view(np.uint32))
    sh = np.zeros_like(a)
    for r in range(4):
        dev.run_raytracer(W, H, 8, seed=7, shard_index=r, shard_count=4, shard_block=8 * W, out=sh)
    assert np.array_equal(sh.view(np.uint32), a.view(np.uint32))
    b, _ = dev.run_raytracer(W, H, 8, seed=7, sort_mode=gpu.RT_SORT_OFF, max_paths=3_000_000)
    assert np.array_equal(b.view(np.uint32), a.view(np.uint32))
    assert np.isfinite(a).all() and a.mean() > 1e-4
    # a second and a third scene in the same process, their FIRST render being a sorted wavefront render: the workspace
    # set-up of a new scene must be ordered with its own stream (regression: a null-stream memset of the queue counters
    # raced with the first launches on the scene's non-blocking stream)
    dev2 = gpu.DeviceScene(sc)
    a2, _ = dev2.run_raytracer(W, H, 8, seed=7)
    assert np.array_equal(a2.view(np.uint32), a.view(np.uint32))
    dev.close()
    dev3 = gpu.DeviceScene(sc)
    a3, _ = dev3.run_raytracer(W, H, 8, seed=7, counters=True)
    assert np.array_equal(a3.view(np.uint32), a.view(np.uint32))
    dev2.close()
    dev3.close()


def _cmp_render(gpu, oracle, sc, W=40, H=36, SPP=5, seed=3):
    dev, orc = gpu.DeviceScene(sc), oracle.OracleScene(sc)
    try:
        # the wavefront pipeline with primary rays through wf_extend (per lane) and through wf_extend_packet (64-ray packets),
        # then the persistent megakernel: one image, one set of event counts
        o, os_ = orc.run_raytracer(W, H, SPP, seed=seed)
        for kw, pkt in (({}, gpu.RT_PACKET_OFF), ({}, gpu.RT_PACKET_ON), ({"megakernel": True}, gpu.RT_PACKET_AUTO)):
            g, gs = dev.run_raytracer(W, H, SPP, seed=seed, counters=True, packet_mode=pkt, **kw)
            assert np.array_equal(g.view(np.uint32), o.view(np.uint32)), (kw, pkt)
            for k in ("casts", "nodes_visited", "box_tests", "tri_tests", "shaded_hits", "texel_fetches", "light_tri_tests"):
                assert gs[k] == os_[k], (k, kw, pkt)
        rays = random_rays(sc, 4000, seed=17) if sc.n_triangles else np.random.default_rng(1).normal(size=(100, 6)).astype(np.float32)
        gp, gb = dev.cast_rays(rays)
        op, ob = orc.cast_rays(rays)
        assert np.array_equal(gp, op) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
        return g
    finally:
        dev.close()
        orc.close()


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_small_ray_depths(gpu, oracle, sg, depth):
    sc = sg.room_scene(300, seed=5, n_lights=3, n_materials=5, tex_size=8, n_tex_sets=2, alpha_fraction=0.3)
    sc.ray_depth = depth
    _cmp_render(gpu, oracle, sc)


def test_empty_scene_renders_background(gpu, oracle, sg):
    """No objects at all: BVH::build returns root = NO_CHILD (bvh.h:373-376); every sample is the white environment."""
    sc = sg.room_scene(0, seed=1, n_lights=0, open_room=True)
    assert sc.n_triangles == 0
    fb = _cmp_render(gpu, oracle, sc)
    assert np.array_equal(fb, np.ones_like(fb))


def test_big_leaves_and_duplicate_geometry(gpu, oracle, sg):
    """Many triangles with identical centroids: the SAH sweep finds no split (bvh.h:299-312), so leaves hold dozens of
    triangles (walked with the per-triangle flags instead of the cooperative <= 8 path) and equal-t hits occur: the
    first triangle in leaf order must win (bvh.h:132)."""
    sc = sg.boxes_scene(n_boxes=5, seed=8, n_lights=2)
    tri = sc.positions[20:21]
    dup = np.repeat(tri, 40, axis=0)
    sc.positions = np.concatenate([sc.positions, dup, dup * np.float32(1.0)], axis=0).astype(np.float32)
    n = sc.positions.shape[0]
    sc.material_ids = np.concatenate([sc.material_ids, np.full(80, 3, dtype=np.uint32)])
    sc.texcoords = np.zeros((n, 3, 2), dtype=np.float32)
    sc.tangents = np.zeros((n, 3, 3), dtype=np.float32)
    sc.tangents[..., 0] = 1
    dev = gpu.DeviceScene(sc)
    info = dev.bvh_info(0)
    leaf_sizes = (info["nodes"][:, 9] - info["nodes"][:, 8])[info["nodes"][:, 6] == 0xFFFFFFFF]
    assert leaf_sizes.max() > 8, "the test needs a leaf beyond the cooperative limit"
    dev.close()
    _cmp_render(gpu, oracle, sc)


def test_single_texel_textures_skip_gamma(gpu, oracle, sg):
    """1x1 loaded textures take Texture::sample's fast path and are returned WITHOUT gamma (geometry.h:548-550)."""
    sc = sg.room_scene(200, seed=9, n_lights=2, n_materials=4, tex_size=0)
    sc.textures = [np.array([[[200, 100, 50, 255]]], dtype=np.uint8), np.array([[[128, 128, 255, 255]]], dtype=np.uint8), np.array([[[0, 180, 90, 255]]], dtype=np.uint8)]
    for m in sc.materials[2:]:
        m.color_tex, m.normal_tex, m.metallic_roughness_tex = 0, 1, 2
    sc.texcoords = np.random.default_rng(2).uniform(-3, 3, size=sc.texcoords.shape).astype(np.float32)
    _cmp_render(gpu, oracle, sc)


@pytest.mark.parametrize("case", ["huge_scene", "tiny_camera_component", "tiny_box_coordinate"])
def test_render_outside_fast_division_range(gpu, oracle, sg, case):
    """The wavefront kernel's wave-uniform switch between the straight-line fast node step and the general step
    (reference IEEE division): a scene scaled by 2^41 (every box coordinate beyond 2^40), a camera whose position has a
    component of 1e-13 (every primary ray guarded, later bounces fast), and a scene with one box coordinate of 1e-13
    (per-scene guarantee void: all rays take the exact path)."""
    sc = sg.room_scene(300, seed=41, n_lights=3, n_materials=5, tex_size=8, n_tex_sets=2)
    if case == "huge_scene":
        k = np.float32(2.0**41)
        sc.positions = (sc.positions * k).astype(np.float32)
        sc.camera.position = (np.asarray(sc.camera.position, dtype=np.float32) * k).astype(np.float32)
    elif case == "tiny_camera_component":
        pos = np.asarray(sc.camera.position, dtype=np.float32).copy()
        pos[2] = np.float32(1e-13)
        sc.camera.position = pos
    else:
        sc.positions = sc.positions.copy()
        sc.positions[20, 0, 1] = np.float32(1e-13)
    _cmp_render(gpu, oracle, sc, W=40, H=32, SPP=4)


def _tiny_scene(sg, n_tris, seed):
    """n_tris triangles in front of a camera at the origin looking down -z; the last one is emissive."""
    rng = np.random.default_rng(seed)
    c = rng.uniform([-2.0, -1.5, -7.0], [2.0, 1.5, -4.0], size=(n_tris, 1, 3))
    pos = (c + rng.uniform(-2.0, 2.0, size=(n_tris, 3, 3)) * np.array([1.0, 1.0, 0.3])).astype(np.float32)
    mats = [sg.Material(color=(0.7, 0.6, 0.5, 1.0), roughness=0.6, metallic=0.2), sg.Material(color=(1, 1, 1, 1), emission=(1.0, 0.9, 0.8), emissive_strength=9.0, roughness=1.0, metallic=0.0)]
    ids = np.zeros(n_tris, dtype=np.uint32)
    ids[-1] = 1
    tang = np.tile(np.array([1, 0, 0], dtype=np.float32), (n_tris, 3, 1))
    return sg.Scene(positions=pos, normals=None, texcoords=rng.uniform(0, 1, size=(n_tris, 3, 2)).astype(np.float32), tangents=tang, material_ids=ids,
                    materials=mats, textures=[], camera=sg.look_camera((0.0, 0.0, 0.0), yaw_deg=0.0, yfov=0.9, aspect=32 / 24))


@pytest.mark.parametrize("n_tris", [1, 2, 3, 5, 9])
def test_tiny_scenes_root_leaf(gpu, oracle, sg, n_tris):
    """Scenes so small that a BVH root is itself a leaf (bvh.h:336-346) or the tree is one or two levels deep, with one
    emissive triangle: the wavefront kernel starts on a leaf reference and the light BVH is a single leaf."""
    g = _cmp_render(gpu, oracle, _tiny_scene(sg, n_tris, 100 + n_tris), W=32, H=24, SPP=4)
    assert (g != 1.0).any()  # something was hit


def test_degenerate_triangles(gpu, oracle, sg):
    """Zero-area triangles (two equal vertices, three collinear vertices, all three equal) and needle-thin ones: the
    reference divides by a zero determinant / normalises a zero normal for them; whatever NaN or infinity that produces
    must come out the same way on the GPU (hits, counters, framebuffer after sanitize_nans)."""
    sc = sg.room_scene(500, seed=77, n_lights=3, n_materials=6, tex_size=8, n_tex_sets=2)
    pos = sc.positions.copy()
    rng = np.random.default_rng(5)
    idx = rng.choice(np.arange(12, 500), size=60, replace=False)  # keep the room walls and lights intact
    for n, t in enumerate(idx):
        kind = n % 4
        if kind == 0:
            pos[t, 1] = pos[t, 0]  # two equal vertices
        elif kind == 1:
            pos[t, 2] = pos[t, 0] + (pos[t, 1] - pos[t, 0]) * np.float32(0.25)  # collinear (up to rounding)
        elif kind == 2:
            pos[t, 1] = pos[t, 0]
            pos[t, 2] = pos[t, 0]  # a point
        else:
            pos[t, 2] = pos[t, 1] + np.float32(1e-7) * (pos[t, 2] - pos[t, 1])  # needle
    sc.positions = pos
    _cmp_render(gpu, oracle, sc, W=48, H=40, SPP=6)


def test_texture_views_mixed_sizes_and_sharing(gpu, oracle, sg):
    """rt_create stores texels tiled and interleaves the equally sized textures of a material (DevTexture views); the
    sampled values must not notice: odd sizes that are no multiple of a tile, a material whose slots differ in size,
    1x1 slots next to real ones, one texture used by several materials in different combinations, an emissive map,
    texcoords that wrap many times."""
    rng = np.random.default_rng(21)
    sc = sg.room_scene(400, seed=13, n_lights=3, n_materials=8, tex_size=0)

    def tex(w, h):
        t = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        t[..., 3] = 255
        return t

    sc.textures = [tex(5, 3), tex(5, 3), tex(5, 3), tex(9, 7), tex(16, 4), tex(1, 1), tex(9, 7), tex(33, 17), tex(33, 17)]
    combos = [(0, 1, 2, -1), (0, 3, 2, -1), (3, 6, 4, 5), (7, 8, 5, 7), (5, 5, 5, -1), (8, -1, 7, 3), (4, 4, 4, 4), (0, 1, 2, 0)]
    for m, (c, n, mr, e) in zip(sc.materials[2:], combos):
        m.color_tex, m.normal_tex, m.metallic_roughness_tex, m.emissive_tex = c, n, mr, e
        if e >= 0:
            m.emission = (0.3, 0.2, 0.1)
    sc.texcoords = rng.uniform(-7, 7, size=sc.texcoords.shape).astype(np.float32)
    # exact texel-grid and wrap boundaries: u, v multiples of 1/w, integers, and values that round up to 1.0f
    sc.texcoords[:40] = rng.integers(-3, 4, size=(40, 3, 2)).astype(np.float32) / np.float32(5)
    sc.texcoords[40:60] = np.float32(-1e-9)
    _cmp_render(gpu, oracle, sc, W=48, H=40, SPP=6)


def test_shard_union_equals_single(pairs, gpu):
    """Image-row tiles sharded over G ranks: the union of the shards is bit-identical to one GPU."""
    dev, _, _ = pairs["room_plain"]
    W, H, SPP = 64, 48, 4
    full, _ = dev.run_raytracer(W, H, SPP, seed=9)
    for G in (2, 3, 8):
        fb = np.full((H, W, 3), -1.0, dtype=np.float32)
        for r in range(G):
            dev.run_raytracer(W, H, SPP, seed=9, shard_index=r, shard_count=G, shard_block=256, out=fb)
        assert np.array_equal(fb.view(np.uint32), full.view(np.uint32)), f"G={G}"


def test_render_is_deterministic(pairs, gpu):
    dev, _, _ = pairs["room_textured"]
    a, _ = dev.run_raytracer(40, 40, 5, seed=3)
    b, _ = dev.run_raytracer(40, 40, 5, seed=3)
    c, _ = dev.run_raytracer(40, 40, 5, seed=4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert not np.array_equal(a, c)


def test_ray_depth_zero_is_noop(gpu, sg):
    sc = sg.boxes_scene(n_boxes=2, seed=1)
    sc.ray_depth = 0
    dev = gpu.DeviceScene(sc)
    fb = np.full((8, 8, 3), 7.0, dtype=np.float32)
    dev.run_raytracer(8, 8, 2, out=fb)
    assert (fb == 7.0).all()  # raytracer.h:630-631
    dev.close()


def test_errors_are_reported_not_thrown(gpu, sg):
    sc = sg.boxes_scene(n_boxes=2, seed=1)
    dev = gpu.DeviceScene(sc)
    with pytest.raises(gpu.RtError):
        dev.run_raytracer(0, 8, 1)
    with pytest.raises(gpu.RtError):
        dev.run_raytracer(8, 8, 1, rng_mode=gpu.RT_RNG_REFERENCE, shard_count=2, shard_block=100)
    dev.close()
    # non-finite vertex positions never reach a builder or a kernel (every build mode)
    for bad in (np.nan, np.inf, -np.inf):
        sc2 = sg.boxes_scene(n_boxes=2, seed=1)
        sc2.positions = sc2.positions.copy()
        sc2.positions[3, 1, 2] = bad
        for kw in ({}, {"device_bvh": True}, {"device_bvh": True, "wide": True}, {"wide": True}):
            with pytest.raises(gpu.RtError) as e:
                gpu.DeviceScene(sc2, **kw)
            assert e.value.code == 1 and "non-finite vertex position (triangle 3)" in str(e.value)  # RT_ERR_INVALID_ARG


def test_cli_end_to_end(gpu, oracle, sg, tmp_path):
    """run.sh <gltf> <W> <H> <SPP> <out.ppm> (reference run.sh:2 / main.cpp:17-25): loader -> HIP render -> film ->
    PPM equals the oracle's PPM for the same seed; argument errors keep the reference's message and exit code."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sc = sg.room_scene(700, seed=91, n_lights=3, n_materials=6, tex_size=8, n_tex_sets=2, alpha_fraction=0.2)
    path = sg.write_gltf(sc, str(tmp_path / "cli.gltf"))
    out = tmp_path / "sub" / "cli.ppm"
    env = dict(os.environ, RT_SEED="5", RT_RNG_MODE="device")
    subprocess.check_call([os.path.join(root, "run.sh"), path, "48", "40", "3", str(out)], env=env)
    ls = gpu.parse_gltf_scene(path, 48 / 40)
    fb, _ = oracle.OracleScene(ls).run_raytracer(48, 40, 3, rng_mode=gpu.RT_RNG_DEVICE, seed=5)
    assert np.array_equal(oracle.read_ppm(str(out)), oracle.tonemap(fb))
    r = subprocess.run([os.path.join(root, "run.sh"), path, "48"], capture_output=True, text=True)
    assert r.returncode == 1 and "Too few arguments: expected 6, got 2" in r.stderr
    r = subprocess.run([os.path.join(root, "run.sh"), str(tmp_path / "nope.gltf"), "8", "8", "1", str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.strip() != ""


def test_reference_rng_mode_equals_reference_binary_golden(gpu, sg, oracle, tmp_path):
    """GPU in RT_RNG_REFERENCE mode against the golden PPM the unmodified reference binary produced (tests/golden,
    made by oracle/make_golden.py from /root/reference's own sources): the same random stream, the same sinf / cosf,
    the same arithmetic -> the same bytes, without the oracle in between."""
    import os

    from conftest import golden_scene_specs, make_scene

    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name in ("room_plain", "room_textured", "room_manylights", "boxes", "open_nolight"):
        sc = make_scene(sg, golden_scene_specs()[name])
        path = sg.write_gltf(sc, str(tmp_path / (name + ".gltf")))
        ls = gpu.parse_gltf_scene(path, 64 / 48)
        dev = gpu.DeviceScene(ls)
        fb, _ = dev.run_raytracer(64, 48, 4, rng_mode=gpu.RT_RNG_REFERENCE)
        img = gpu.tonemap(fb)
        ref = oracle.read_ppm(os.path.join(gold_dir, f"{name}_64x48x4.ppm"))
        differing = int((img != ref).any(axis=2).sum())
        assert differing == 0, f"{name}: {differing} of {64 * 48} pixels differ from the reference binary's PPM"
        dev.close()


def test_feature_gltf_through_loader_matches_oracle_and_reference(gpu, oracle):
    """The hand-built loader-feature glTF (tests/golden/features: scene selection, matrix + TRS nodes, strips, u8/u16/u32
    indices, missing attributes, emissive texture + strength, mixed texture sizes) through the C++ loader onto the GPU:
    bit-exact against the oracle in device-RNG mode, and in reference-RNG mode byte-identical to the PPM the unmodified
    reference binary produced."""
    import os

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ls = gpu.parse_gltf_scene(os.path.join(gold, "features", "features.gltf"), 64 / 48)
    dev, orc = gpu.DeviceScene(ls), oracle.OracleScene(ls)
    g, gs = dev.run_raytracer(64, 48, 6, seed=17, counters=True)
    o, os_ = orc.run_raytracer(64, 48, 6, seed=17)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
    for k in ("casts", "nodes_visited", "tri_tests", "shaded_hits", "texel_fetches", "light_tri_tests"):
        assert gs[k] == os_[k], k
    img, _ = dev.run_raytracer_rgb8(64, 48, 4, rng_mode=gpu.RT_RNG_REFERENCE)
    ref = oracle.read_ppm(os.path.join(gold, "features_64x48x4.ppm"))
    differing = int((img != ref).any(axis=2).sum())
    assert differing == 0, f"{differing} of {64 * 48} pixels differ from the reference binary's PPM"
    dev.close()
    orc.close()


# ------------------------------------------------------------------------------------------------ device film (8f-3)
def _film_edge_values():
    rng = np.random.default_rng(12)
    thr, _ = None, None
    vals = [rng.uniform(0, 4, 200000), rng.uniform(0, 0.02, 20000), 10.0 ** rng.uniform(-44, 38, 50000), -(10.0 ** rng.uniform(-44, 38, 2000)),
            [0.0, -0.0, 1.0, 0.18, 1e6, 3.4e38, np.inf, -np.inf, np.nan, 1e-45, -1e-45, -0.012345679, -0.0123456]]
    with np.errstate(over="ignore"):
        return np.concatenate([np.asarray(v, dtype=np.float64) for v in vals]).astype(np.float32)


def test_device_film_matches_oracle_film(pairs, gpu, oracle):
    """rt_film_rgb8 (ACES on the device + verified threshold table) == the oracle's film (image.h:49-82 with libm powf),
    byte for byte: random radiances, all exponents, negatives, NaN, infinities, and every float within 3 ulps of every
    one of the 255 level thresholds mapped back through the ACES curve's neighbourhood."""
    dev, _, _ = pairs["room_plain"]
    x = _film_edge_values()
    # radiances whose ACES value lands next to a threshold: bisect x for each threshold, then take its float neighbours
    thr, special = gpu.film_table()
    a, b, c, d, e = (np.float32(v) for v in (2.51, 0.03, 2.43, 0.59, 0.14))
    lo, hi = np.zeros(256, dtype=np.float32), np.full(256, 64.0, dtype=np.float32)
    for _ in range(60):
        mid = ((lo.astype(np.float64) + hi) / 2).astype(np.float32)
        y = (mid * (a * mid + b)) / (mid * (c * mid + d) + e)
        below = y < thr
        lo, hi = np.where(below, mid, lo), np.where(below, hi, mid)
    near = (hi.view(np.uint32)[:, None].astype(np.int64) + np.arange(-6, 7)[None, :]).clip(0, 0x7F7FFFFF).astype(np.uint32).view(np.float32).ravel()
    x = np.concatenate([x, near])
    x = np.concatenate([x, np.zeros((-x.size) % 3, dtype=np.float32)]).reshape(-1, 1, 3)
    got = dev.film_rgb8(x)
    want = oracle.tonemap(x)
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} values differ"
    assert np.array_equal(got, gpu.tonemap(x))  # and the product's own host film
    assert got.min() == 0 and got.max() == 255


@pytest.mark.parametrize("name", ["room_textured", "boxes"])
def test_render_rgb8_equals_film_of_float_render(pairs, gpu, oracle, name):
    """rt_render_rgb8 == oracle film of the oracle framebuffer (hence == the PPM bytes the CPU path would write),
    on one GPU and as the union of shards; pixels of other shards are not touched."""
    dev, orc, _ = pairs[name]
    W, H, SPP = 64, 48, 4
    ofb, _ = orc.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_DEVICE, seed=21)
    want = oracle.tonemap(ofb)
    img, st = dev.run_raytracer_rgb8(W, H, SPP, seed=21)
    assert np.array_equal(img, want)
    assert st["samples"] == W * H * SPP
    for G in (2, 5):
        acc = np.full((H, W, 3), 77, dtype=np.uint8)
        for r in range(G):
            before = acc.copy()
            dev.run_raytracer_rgb8(W, H, SPP, seed=21, shard_index=r, shard_count=G, shard_block=256, out=acc)
            mine = np.zeros(W * H, dtype=bool)
            for b0 in range(r * 256, W * H, G * 256):
                mine[b0 : b0 + 256] = True
            assert np.array_equal(acc.reshape(-1, 3)[~mine], before.reshape(-1, 3)[~mine])
        assert np.array_equal(acc, want), f"G={G}"
    # reference-RNG mode through the megakernel takes the same film
    rfb, _ = dev.run_raytracer(W, H, 2, rng_mode=gpu.RT_RNG_REFERENCE)
    rimg, _ = dev.run_raytracer_rgb8(W, H, 2, rng_mode=gpu.RT_RNG_REFERENCE)
    assert np.array_equal(rimg, oracle.tonemap(rfb))


def test_render_rgb8_device_destination(pairs, gpu, oracle):
    """RT_FLAG_DEVICE_FB with rt_render_rgb8: the image stays in HBM (what the multi-GPU gather consumes)."""
    import torch

    dev, orc, _ = pairs["room_plain"]
    W, H, SPP = 40, 24, 3
    t = torch.full((H * W * 3,), 9, dtype=torch.uint8, device="cuda")
    dev.run_raytracer_rgb8(W, H, SPP, seed=2, device_rgb8=t.data_ptr())
    torch.cuda.synchronize()
    ofb, _ = orc.run_raytracer(W, H, SPP, rng_mode=gpu.RT_RNG_DEVICE, seed=2)
    assert np.array_equal(t.cpu().numpy().reshape(H, W, 3), oracle.tonemap(ofb))


def test_config4_shape_1000_spp_on_the_bench_scene(gpu, oracle, sg):
    """BASELINE config 4 on one GPU: S-sponza 1000x1000 at 1000 SPP = 10^9 samples, rendered in 8 sample passes of 125 M
    paths (rt_params.max_paths = 0: 128 M per pass) whose partial sums must continue in sample order (raytracer.h:621-626). Two 256-pixel spans of the image are
    recomputed by the oracle at the full 1000 SPP (bit-exact), and the union of the 8 interleaved shards the 8 ranks of that
    configuration would render equals the single render (same blocks, same bytes the RCCL gather would move)."""
    W = H = 1000
    SPP = 1000
    sc = sg.room_scene(262144, seed=0x5EED5EED, tex_size=256, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0,
                       alpha_fraction=0.02, offset=0.15, camera=sg.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
    dev = gpu.DeviceScene(sc)
    orc = oracle.OracleScene(sc)
    try:
        img, st = dev.run_raytracer_rgb8(W, H, SPP, seed=0xC4)
        assert st["samples"] == W * H * SPP and st["passes"] == 8 and st["dominant_launches"] >= 8 * 8 - 8  # 8 passes x (up to) 8 bounces
        fb, _ = dev.run_raytracer(W, H, SPP, seed=0xC4)
        assert np.array_equal(img, gpu.tonemap(fb))
        n_spans = (W * H + 255) // 256
        for span in (1234, 3000):
            ofb = np.full((H, W, 3), -1.0, dtype=np.float32)
            orc.run_raytracer(W, H, SPP, seed=0xC4, shard_index=span, shard_count=n_spans, shard_block=256, out=ofb)
            mine = (ofb.reshape(-1, 3)[:, 0] != -1.0)
            assert int(mine.sum()) == 256
            assert np.array_equal(fb.reshape(-1, 3)[mine].view(np.uint32), ofb.reshape(-1, 3)[mine].view(np.uint32)), span
        acc = np.zeros_like(img)
        for r in range(8):
            dev.run_raytracer_rgb8(W, H, SPP, seed=0xC4, shard_index=r, shard_count=8, shard_block=8 * W, out=acc)
        assert np.array_equal(acc, img)
    finally:
        dev.close()
        orc.close()

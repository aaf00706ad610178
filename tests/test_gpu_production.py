"""Production traversal modes against the CPU oracle.

The parity mode reproduces the reference's traversal exactly (bvh.h:195-235: near-local pruning, reference topology) and is
pinned elsewhere bit for bit. The production modes do LESS work than the reference:
  * RT_FLAG_GLOBAL_BEST / RT_CAST_*_GLOBAL : every box is culled against the global best hit so far;
  * RT_BUILD_WIDE                          : 8-wide nodes with quantised child boxes (tests further below, when built).
Their contract is checked against the ORACLE (never against another GPU scene): the closest hit's `t` must be bit-equal on
every ray; hit indices may differ only where two triangles share that exact `t` (a tie the reference resolves by its own
traversal order); the mismatch count is asserted, not printed. The event counters must show the mode really visits fewer nodes.
"""
import numpy as np
import pytest

from conftest import random_rays

pytestmark = pytest.mark.gpu

FIXTURES = ["room_plain", "room_textured", "open_nolight", "boxes", "room_manylights"]


def compare_hits_with_oracle(op, ob, gp, gb, what):
    """`t` bit-equal on every ray; an index mismatch is legal only as an exact tie (same t bits, both hit). Returns the tie count."""
    ot, gt = ob[:, 2].view(np.uint32), gb[:, 2].view(np.uint32)
    miss_o, miss_g = op == 0xFFFFFFFF, gp == 0xFFFFFFFF
    assert np.array_equal(miss_o, miss_g), f"{what}: {int((miss_o != miss_g).sum())} rays hit on one side only"
    bad_t = ot != gt
    assert not bad_t.any(), f"{what}: t differs from the oracle on {int(bad_t.sum())} of {len(op)} rays (first: ray {int(np.flatnonzero(bad_t)[0])})"
    ties = (op != gp) & ~miss_o
    # where the index agrees, the barycentrics are the same triangle test on the same operands: bit-equal too
    same = ~ties
    assert np.array_equal(gb[same].view(np.uint32), ob[same].view(np.uint32)), f"{what}: b/c differ on rays with the oracle's own triangle"
    return int(ties.sum())


@pytest.fixture(scope="module")
def pairs(gpu, oracle, scenes):
    out = {}
    for name, sc in scenes.items():
        out[name] = (gpu.DeviceScene(sc), oracle.OracleScene(sc), sc)
    yield out
    for d, o, _ in out.values():
        d.close()
        o.close()


def _camera_rays(sc, n, seed):
    """Rays from the camera position (what primary rays look like: one origin, a packet-friendly fan of directions)."""
    rng = np.random.default_rng(seed)
    cam = sc.camera
    f, r, u = np.asarray(cam.forward, np.float32), np.asarray(cam.right, np.float32), np.asarray(cam.up, np.float32)
    a = rng.uniform(-0.5, 0.5, size=(n, 2)).astype(np.float32)
    a = a[np.lexsort((a[:, 0], np.floor(a[:, 1] * 64)))]  # scanline-ish order: 64 consecutive rays are neighbours
    d = f[None, :] + a[:, :1] * r[None, :] + a[:, 1:] * u[None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(np.asarray(cam.position, np.float32), (n, 1))
    return np.concatenate([o, d.astype(np.float32)], axis=1).astype(np.float32)


@pytest.mark.parametrize("name", FIXTURES)
def test_wavefront_probe_in_parity_mode_is_the_oracle(pairs, gpu, name):
    """rt_cast_rays_ex through wf_extend / wf_extend_packet in the reference's traversal order: every field bit-equal, and
    the kernels' event counters equal the per-lane probe's semantics (nodes and triangle tests are the oracle's)."""
    dev, orc, sc = pairs[name]
    rays = np.concatenate([random_rays(sc, 6000, seed=7), _camera_rays(sc, 4096 + 37, seed=8)])
    op, ob = orc.cast_rays(rays)
    for mode in (gpu.RT_CAST_EXTEND, gpu.RT_CAST_PACKET):
        gp, gb, st = dev.cast_rays_ex(rays, mode)
        assert np.array_equal(gp, op), (name, mode, int((gp != op).sum()))
        assert np.array_equal(gb.view(np.uint32), ob.view(np.uint32)), (name, mode)
        assert st["casts"] == len(rays) and st["nodes_visited"] > 0


@pytest.mark.parametrize("name", FIXTURES)
def test_global_best_hits_equal_the_oracle(pairs, gpu, name):
    dev, orc, sc = pairs[name]
    rays = np.concatenate([random_rays(sc, 20000, seed=101), _camera_rays(sc, 8192, seed=102)])
    op, ob = orc.cast_rays(rays)
    _, _, st_ref = dev.cast_rays_ex(rays, gpu.RT_CAST_EXTEND)
    for mode in (gpu.RT_CAST_EXTEND_GLOBAL, gpu.RT_CAST_PACKET_GLOBAL):
        gp, gb, st = dev.cast_rays_ex(rays, mode)
        ties = compare_hits_with_oracle(op, ob, gp, gb, f"{name} mode {mode}")
        # random triangles never tie exactly; the boxes scene has shared edges (two triangles of a face, same t along the diagonal)
        assert ties <= (40 if name == "boxes" else 0), (name, mode, ties)
        # a subset of the reference's visits, and a real saving
        assert st["nodes_visited"] <= st_ref["nodes_visited"] and st["tri_tests"] <= st_ref["tri_tests"], (name, mode, st, st_ref)
    print(f"{name}: nodes visited global-best / reference = {st['nodes_visited']} / {st_ref['nodes_visited']} = {st['nodes_visited'] / st_ref['nodes_visited']:.4f}")


@pytest.mark.parametrize("name", FIXTURES)
def test_global_best_render_matches_oracle(pairs, gpu, name):
    """The whole pipeline with global-best traversal: same closest hits -> the oracle's framebuffer (1e-5 relative, the
    stated tolerance; a tie resolved differently would show up here as a different material / normal)."""
    dev, orc, _ = pairs[name]
    W, H, SPP = 48, 40, 6
    gfb, gst = dev.run_raytracer(W, H, SPP, seed=5, global_best=True, counters=True)
    ofb, ost = orc.run_raytracer(W, H, SPP, seed=5)
    rel = np.abs(gfb - ofb) / np.maximum(np.abs(ofb), 1e-6)
    assert rel.max() <= 1e-5, (name, float(rel.max()), int((rel > 1e-5).any(axis=2).sum()))
    assert gst["casts"] == ost["casts"] and gst["shaded_hits"] == ost["shaded_hits"]  # same paths ...
    assert gst["nodes_visited"] <= ost["nodes_visited"]  # ... found with no more node visits


def test_global_best_on_the_bench_scene(gpu, oracle, sg):
    """S-sponza at full size (BASELINE config 3's scene): 60 000 rays and the 1000 x 1000 x 1 SPP framebuffer against the
    oracle; nodes / triangle tests per cast of both traversals are reported in the assertion messages' terms."""
    sc = sg.room_scene(262144, seed=0x5EED5EED, tex_size=64, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0,
                       alpha_fraction=0.02, offset=0.15, camera=sg.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
    dev, orc = gpu.DeviceScene(sc), oracle.OracleScene(sc)
    try:
        rays = random_rays(sc, 60000, seed=4242)
        op, ob = orc.cast_rays(rays)
        _, _, st_ref = dev.cast_rays_ex(rays, gpu.RT_CAST_EXTEND)
        gp, gb, st = dev.cast_rays_ex(rays, gpu.RT_CAST_EXTEND_GLOBAL)
        assert compare_hits_with_oracle(op, ob, gp, gb, "S-sponza, global best") == 0
        assert st["nodes_visited"] <= st_ref["nodes_visited"], (st["nodes_visited"] / len(rays), st_ref["nodes_visited"] / len(rays))
        W = H = 1000
        gfb, gst = dev.run_raytracer(W, H, 1, seed=0x5EED5EED, global_best=True, counters=True)
        ofb, ost = orc.run_raytracer(W, H, 1, seed=0x5EED5EED)
        diff = (gfb.view(np.uint32) != ofb.view(np.uint32)).any(axis=2)
        assert not diff.any(), f"{int(diff.sum())} of 10^6 pixels differ from the oracle"
        assert gst["casts"] == ost["casts"]
        print(f"S-sponza nodes/cast {gst['nodes_visited'] / gst['casts']:.1f} (oracle {ost['nodes_visited'] / ost['casts']:.1f}), "
              f"triangle tests/cast {gst['tri_tests'] / gst['casts']:.1f} (oracle {ost['tri_tests'] / ost['casts']:.1f})")
    finally:
        dev.close()
        orc.close()

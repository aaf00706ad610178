"""BASELINE config 5 (S-10M: 10^7 random triangles, 2048x2048) as a TESTED configuration: the HIP path through the C-ABI
against the CPU oracle at full scene size, plus the size-independent properties on the full image.

The oracle cannot render 4 M pixels of this scene in seconds, so the pixel comparison takes every 64th 256-pixel span of
the 2048x2048 image (65 536 pixels spread over the whole frame, shard_count = 64) at 1 SPP; hit records are compared on
20 000 random rays; config 5's own depth (256 SPP, 1.07e9 samples) is rendered whole and two of its spans are recomputed by the
oracle at all 256 samples; the rest are properties that need no oracle (wavefront == megakernel, union of 8 shards == single
render, pass splitting). RT_TEST_S10M_TRIANGLES overrides the triangle count for a quicker local run.
"""
import os

import numpy as np
import pytest

from conftest import explain_differing_pixels, random_rays

pytestmark = pytest.mark.gpu

COUNTERS = ("samples", "casts", "nodes_visited", "box_tests", "tri_tests", "shaded_hits", "light_queries", "light_nodes", "light_box_tests",
            "light_tri_tests", "light_hits", "texel_fetches")


def test_s10m_full_size_parity_and_properties(gpu, oracle, sg):
    n = int(float(os.environ.get("RT_TEST_S10M_TRIANGLES", "1e7")))
    W = H = 2048
    sc = sg.room_scene(n, seed=0x5EED5EED, tex_size=256, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0, alpha_fraction=0.02,
                       offset=0.03, camera=sg.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
    dev = gpu.DeviceScene(sc)
    orc = oracle.OracleScene(sc)
    try:
        # the BVH rt_create flattened and uploaded is the oracle's (reference topology), node for node
        for which in (0, 1):
            a, b = dev.bvh_info(which), orc.bvh_info(which)
            assert a["root"] == b["root"] and np.array_equal(a["order"], b["order"]) and np.array_equal(a["nodes"], b["nodes"])
        if n >= 10_000_000:
            assert a["nodes"].shape[0] > 1 and dev.bvh_info(0)["nodes"].shape[0] > 4_000_000
        # every 64th span of the full-size image at 1 SPP: bit-exact radiance + identical event counters
        gfb = np.full((H, W, 3), -1.0, dtype=np.float32)
        ofb = np.full((H, W, 3), -1.0, dtype=np.float32)
        _, gst = dev.run_raytracer(W, H, 1, seed=0x5EED5EED, shard_index=5, shard_count=64, shard_block=256, out=gfb, counters=True)
        _, ost = orc.run_raytracer(W, H, 1, seed=0x5EED5EED, shard_index=5, shard_count=64, shard_block=256, out=ofb)
        assert np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32)), f"{int((gfb != ofb).any(axis=2).sum())} of {W * H // 64} pixels differ"
        assert int((gfb[..., 0] != -1.0).sum()) == W * H // 64
        for k in COUNTERS:
            assert gst[k] == ost[k], (k, gst[k], ost[k])
        assert gst["casts"] > 150_000 and gst["nodes_visited"] / gst["casts"] > 100  # deep traversals really happened
        # 20 000 random rays: bit-exact hit records (primitive index, b, c, t)
        rays = random_rays(sc, 20000, seed=77)
        gp, gb = dev.cast_rays(rays)
        op, ob = orc.cast_rays(rays)
        assert np.array_equal(gp, op), f"{int((gp != op).sum())} hit-index mismatches"
        assert np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
        assert (gp != 0xFFFFFFFF).sum() > 15000
        g = dev.light_pdf(rays)
        o = orc.light_pdf(rays)
        assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
        # ---- the production modes on the same 10^7 triangles, each against the ORACLE: the device LBVH, global-best
        # pruning, and the 8-wide tree collapsed from either binary tree. 100 000 rays: t bit-equal on every ray, index
        # mismatches (exact ties) counted and bounded; every 64th span at 1 SPP against the oracle's pixels.
        rays = random_rays(sc, 100_000, seed=78)
        op, ob = orc.cast_rays(rays)

        def check(what, scene, mode, spans=True):
            gp, gb, st = scene.cast_rays_ex(rays, mode)
            assert np.array_equal(op == 0xFFFFFFFF, gp == 0xFFFFFFFF), what
            bad = ob[:, 2].view(np.uint32) != gb[:, 2].view(np.uint32)
            assert not bad.any(), f"{what}: t differs from the oracle on {int(bad.sum())} of {len(rays)} rays"
            ties = int((op != gp).sum())
            assert ties <= 5, f"{what}: {ties} index mismatches (exact ties) in a random triangle soup"
            msg = f"[S-10M] {what}: {ties} ties / {len(rays)} rays, {st['nodes_visited'] / len(rays):.1f} node visits + {st['tri_tests'] / len(rays):.1f} triangle tests per cast"
            if spans:
                pfb = np.full((H, W, 3), -1.0, dtype=np.float32)
                scene.run_raytracer(W, H, 1, seed=0x5EED5EED, shard_index=5, shard_count=64, shard_block=256, out=pfb, global_best=mode == gpu.RT_CAST_EXTEND_GLOBAL)
                nbad = int((pfb.view(np.uint32) != ofb.view(np.uint32)).any(axis=2).sum())
                assert nbad <= 8, f"{what}: {nbad} of {W * H // 64} pixels differ from the oracle"
                msg += f", {nbad} of {W * H // 64} pixels differ from the oracle"
            print(msg)

        check("reference tree, global-best pruning", dev, gpu.RT_CAST_EXTEND_GLOBAL)
        lb = gpu.DeviceScene(sc, device_bvh=True)
        check("device LBVH", lb, gpu.RT_CAST_EXTEND)
        check("device LBVH, global-best pruning", lb, gpu.RT_CAST_EXTEND_GLOBAL, spans=False)
        t_lb = lb.build_times()
        lb.close()
        # the whole 2048 x 2048 image at the bench's 32 SPP (1.3e8 samples) in parity mode: what the production build is counted against
        pfb32, _ = dev.run_raytracer(W, H, 32, seed=0x5EED5EED)
        for what, kw in (("wide tree from the reference-topology tree", dict(wide=True)), ("wide tree from the device LBVH", dict(wide=True, device_bvh=True))):
            wd = gpu.DeviceScene(sc, **kw)
            check(what, wd, gpu.RT_CAST_EXTEND)
            if kw.get("device_bvh"):  # the benched production build, at the depth its number is quoted at
                wfb32, _ = wd.run_raytracer(W, H, 32, seed=0x5EED5EED)
                bits = (wfb32.view(np.uint32) != pfb32.view(np.uint32)).any(axis=2)
                print(f"[S-10M] production build at 32 SPP: {int(bits.sum())} of {W * H} pixels differ from the parity image in any bit")
                # 1.3e8 samples, ~5e8 casts: measured 1 pixel. Every differing pixel must be explained by the production contract's two cases
                assert int(bits.sum()) <= 4, (int(bits.sum()), np.argwhere(bits)[:8].tolist())
                for rec in explain_differing_pixels(gpu, orc, dev, wd, W, H, 32, 0x5EED5EED, np.argwhere(bits)):
                    print(f"[S-10M]   {rec}")
                del wfb32
            wd.close()
        del pfb32
        # BASELINE config 5 at its own depth: 256 SPP = 1.07e9 samples on one GPU, eight sample passes of 32 SPP (128 M paths each) whose partial sums
        # continue in sample order (raytracer.h:621-626); two 256-pixel spans recomputed by the oracle at the full 256 SPP must match bit for bit
        if n >= 10_000_000:
            fb256, st256 = dev.run_raytracer(W, H, 256, seed=0xC5)
            assert st256["samples"] == W * H * 256 and st256["passes"] == 8
            n_spans = W * H // 256
            for span in (4321, 12345):
                sfb = np.full((H, W, 3), -1.0, dtype=np.float32)
                orc.run_raytracer(W, H, 256, seed=0xC5, shard_index=span, shard_count=n_spans, shard_block=256, out=sfb)
                mine = sfb.reshape(-1, 3)[:, 0] != -1.0
                assert int(mine.sum()) == 256
                assert np.array_equal(fb256.reshape(-1, 3)[mine].view(np.uint32), sfb.reshape(-1, 3)[mine].view(np.uint32)), f"config 5 at 256 SPP: span {span} differs from the oracle"
            print(f"[S-10M] config 5 (2048x2048x256) on one GPU: {st256['passes']} passes, {W * H * 256 / st256['kernel_ms'] / 1e3:.1f} Msamples/s device time; 2 spans x 256 SPP bit-equal to the oracle")
            del fb256
        t_ref = dev.build_times()
        print(f"[S-10M] host reference-topology build {t_ref['build_ms'] / 1e3:.2f} s; device LBVH build {t_lb['build_ms']:.1f} ms (+ upload {t_lb['upload_ms']:.0f} ms)")
    finally:
        orc.close()
    # ---- size-independent properties on the whole 2048x2048 image (4 M pixels x 2 SPP = 8 M paths)
    a, ast = dev.run_raytracer(W, H, 2, seed=11, counters=True)
    assert np.isfinite(a).all() and a.mean() > 1e-4
    m, mst = dev.run_raytracer(W, H, 2, seed=11, megakernel=True, counters=True)
    assert np.array_equal(a.view(np.uint32), m.view(np.uint32)), "wavefront pipeline != persistent megakernel"
    for k in COUNTERS:
        assert ast[k] == mst[k], k
    # primary rays as 64-ray packets (wf_extend_packet) even at 2 SPP: incoherent packets, same hits
    pk, pst = dev.run_raytracer(W, H, 2, seed=11, counters=True, packet_mode=gpu.RT_PACKET_ON)
    assert pst["packet_passes"] == pst["passes"] >= 1 and 100 <= pst["packet_lanes_x100"] <= 6400  # the census reaches the host (r03: wiped by the ticket reset)
    assert np.array_equal(pk.view(np.uint32), a.view(np.uint32)), "packet traversal of the primary rays != per-lane traversal"
    for k in COUNTERS:
        assert pst[k] == ast[k], k
    sh = np.zeros_like(a)
    for r in range(8):
        dev.run_raytracer(W, H, 2, seed=11, shard_index=r, shard_count=8, shard_block=8 * W, out=sh)
    assert np.array_equal(sh.view(np.uint32), a.view(np.uint32)), "union of 8 shards != single render"
    b, _ = dev.run_raytracer(W, H, 2, seed=11, max_paths=3_000_000)  # several pixel tiles x sample passes
    assert np.array_equal(b.view(np.uint32), a.view(np.uint32))
    img, _ = dev.run_raytracer_rgb8(W, H, 2, seed=11)
    assert np.array_equal(img, gpu.tonemap(a))
    dev.close()

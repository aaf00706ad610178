#!/usr/bin/env python3
"""tools/soak_parity.py [n_cases] [seed] — development aid: randomized parity soak on the GPU box.

Each case draws a scene of the S-sponza family (triangle count, offsets, lights, materials, textures, alpha share, smooth normals, open or closed
room, camera), an image shape, a sample count, a ray depth and the render knobs (ray-order key, packet mode, paths per pass, shard split), renders it
through librt_amd.so in parity mode and compares with the CPU oracle: framebuffer bit for bit, every event counter. Then the production builds of the
same scene (global-best pruning, device LBVH, the 8-wide tree from either builder) against the oracle's closest hits on random rays: t bit-equal, index
differences on exact ties (and, on binary trees, on overlapping coplanar triangles: <= 4 ulp), the wide tree never farther; pixels of their images beyond 1e-5
relative are counted. Prints one line per case and a summary; exits 1 on the first parity-mode difference, 2 on a production hit outside that contract."""
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

COUNTERS = ("samples", "casts", "nodes_visited", "box_tests", "tri_tests", "shaded_hits", "light_queries", "light_nodes", "light_box_tests",
            "light_tri_tests", "light_hits", "texel_fetches")


def random_rays(rng, lo, hi, n):
    o = rng.uniform(lo, hi, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
    sg = rt.scenegen
    sorts = (rt.RT_SORT_AUTO, rt.RT_SORT_OFF, rt.RT_SORT_CELL_OCTANT, rt.RT_SORT_OCTANT_CELL_CONE, rt.RT_SORT_OCTANT_FINE_CELL_CONE)
    worst_prod = 0
    big = os.environ.get("SOAK_BIG") == "1"  # trees far beyond the caches: 2e5 - 3e6 triangles, every 16th 256-pixel span of a 512 x 512 image against the oracle
    only = int(os.environ.get("SOAK_ONLY", "-1"))  # replay one case of the sequence (same draws), with details on a production difference
    t_start = time.time()
    for case in range(n_cases):
        n_tri = int(rng.choice([0, 1, 7, 60, 500, 4000, 30000]))
        if big:
            n_tri = int(rng.choice([200_000, 1_000_000, 3_000_000]))
        tex = int(rng.choice([0, 0, 4, 32, 64]))
        kw = dict(seed=int(rng.integers(1, 2**31)), offset=float(rng.choice([0.05, 0.3, 1.5])), n_lights=int(rng.choice([0, 1, 3, 16, 40])),
                  light_strength=float(rng.choice([5.0, 20.0])), n_materials=int(rng.choice([1, 5, 64])), tex_size=tex, n_tex_sets=int(rng.choice([1, 3, 16])),
                  alpha_fraction=float(rng.choice([0.0, 0.05, 0.5])), smooth_normals=bool(rng.integers(0, 2)), open_room=bool(rng.integers(0, 4) == 0),
                  camera=sg.look_camera((float(rng.uniform(-18, 18)), float(rng.uniform(1, 15)), float(rng.uniform(-8, 8))), yaw_deg=float(rng.uniform(-180, 180)),
                                        yfov=float(rng.uniform(0.4, 1.4))))
        sc = sg.room_scene(n_tri, **kw)
        W, H, SPP = int(rng.integers(1, 200)), int(rng.integers(1, 160)), int(rng.choice([1, 2, 5, 16, 37]))
        if rng.integers(0, 6) == 0:  # now and then a queue long enough for wf_shade's class-sorted windows (>= 2 M sorted rays)
            W, H, SPP = 640, int(rng.integers(500, 700)), 8
        if big:
            W, H, SPP = 512, 512, 4
        depth = int(rng.choice([1, 2, 3, 8, 8, 12]))
        sc.ray_depth = depth
        if rng.integers(0, 3) == 0:  # texture coordinates far outside [0, 1): the wrap-around texel path (Texture::sample, geometry.h:545-575), negative and huge values
            U, V = float(rng.choice([-3.0, 7.5, 1e3, 1e6])), float(rng.choice([0.0, -0.25, 123.456]))
            sc.texcoords = (sc.texcoords * np.float32(U) + np.float32(V)).astype(np.float32)
        if rng.integers(0, 3) == 0:  # materials at the ends of their ranges: mirror-smooth, fully rough, metal / dielectric, invisible, odd ior, an emissive texture
            for m in sc.materials[2:]:
                r = int(rng.integers(0, 8))
                if r == 0:
                    m.roughness = 0.0
                elif r == 1:
                    m.metallic, m.roughness = 1.0, 0.0
                elif r == 2:
                    m.metallic = 0.0
                elif r == 3:
                    m.color = (m.color[0], m.color[1], m.color[2], 0.0)
                elif r == 4:
                    m.ior = float(rng.choice([1.0, 1.0001, 3.0]))
                elif r == 5 and sc.textures:
                    m.emissive_tex, m.emission, m.emissive_strength = int(rng.integers(0, len(sc.textures))), (1.0, 0.5, 0.25), 2.0
        if kw["open_room"] and tex > 0 and rng.integers(0, 2) == 0:  # an environment map (Scene::bg, scene.h:81-89): any RGBA8 picture serves as the equirect image
            sc.bg_texture = int(rng.integers(0, len(sc.textures)))
        knobs = dict(sort_mode=int(rng.choice(sorts)), packet_mode=int(rng.choice([rt.RT_PACKET_AUTO, rt.RT_PACKET_OFF, rt.RT_PACKET_ON])))
        if rng.integers(0, 3) == 0:
            knobs["max_paths"] = int(rng.choice([1024, 5000, 70000, 1 << 20]))
        seed = int(rng.integers(0, 2**31))
        if only >= 0 and case != only:  # replay: draw what the case would have drawn, render nothing
            if rng.integers(0, 2) == 0:
                rng.choice([2, 3, 8]), rng.choice([1, 64, 256, 1000])
            rng.integers(0, 3), rng.integers(0, 3)
            if rng.integers(0, 4) == 0:
                rng.choice([2, 3, 4]), rng.choice([0, 64, 256, 1000])
            random_rays(rng, np.zeros(3), np.ones(3), 4000)
            continue
        dev = rt.DeviceScene(sc)
        orc = oracle.OracleScene(sc)
        try:
            if big:  # the same shard on both sides: spans 3, 19, 35, ... of the image
                ofb = np.zeros((H, W, 3), dtype=np.float32)
                gfb = np.zeros((H, W, 3), dtype=np.float32)
                _, ost = orc.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_DEVICE, seed=seed, shard_index=3, shard_count=16, shard_block=256, out=ofb)
                _, gst = dev.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_DEVICE, seed=seed, counters=True, shard_index=3, shard_count=16, shard_block=256, out=gfb, **knobs)
            else:
                ofb, ost = orc.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_DEVICE, seed=seed)
                gfb, gst = dev.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_DEVICE, seed=seed, counters=True, **knobs)
            ok = np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32)) and all(gst[k] == ost[k] for k in COUNTERS)
            if ok and big:  # the whole image in both schedules: wavefront pipeline == persistent megakernel
                a, _ = dev.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_DEVICE, seed=seed, **knobs)
                m, _ = dev.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_DEVICE, seed=seed, megakernel=True)
                ok = np.array_equal(a.view(np.uint32), m.view(np.uint32)) and np.array_equal(a.reshape(-1, 3)[gfb.reshape(-1, 3)[:, 0] != 0].view(np.uint32), gfb.reshape(-1, 3)[gfb.reshape(-1, 3)[:, 0] != 0].view(np.uint32))
            if ok and rng.integers(0, 2) == 0 and not big:  # the split render: union of the shards
                cnt = int(rng.choice([2, 3, 8]))
                blk = int(rng.choice([1, 64, 256, 1000]))
                sh = np.zeros_like(ofb)
                for r in range(cnt):
                    dev.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_DEVICE, seed=seed, shard_index=r, shard_count=cnt, shard_block=blk, out=sh)
                ok = np.array_equal(sh.view(np.uint32), ofb.view(np.uint32))
            if ok and rng.integers(0, 3) == 0 and not big:  # the reference's own RNG stream (minstd per 256-pixel span, glibc sinf / cosf): the megakernel path
                w2, h2, s2 = min(W, 96), min(H, 64), min(SPP, 5)
                rfb, _ = dev.run_raytracer(w2, h2, s2, rng_mode=rt.RT_RNG_REFERENCE)
                orf, _ = orc.run_raytracer(w2, h2, s2, rng_mode=rt.RT_RNG_REFERENCE)
                ok = np.array_equal(rfb.view(np.uint32), orf.view(np.uint32))
            if ok and rng.integers(0, 3) == 0 and not big:  # the device film: bytes of the host film applied to the oracle's image
                img, _ = dev.run_raytracer_rgb8(W, H, SPP, rng_mode=rt.RT_RNG_DEVICE, seed=seed, **knobs)
                ok = np.array_equal(img, oracle.tonemap(ofb))
            if ok and rng.integers(0, 4) == 0 and not big:  # the library's own multi-GPU scene: G replicas on GPU 0 over the peer-copy rehearsal transport (csrc/rt_group.cpp)
                G, blk = int(rng.choice([2, 3, 4])), int(rng.choice([0, 64, 256, 1000]))
                grp = rt.DeviceScene(sc, device=[0] * G, build_flags=rt.RT_BUILD_GROUP_COPY)
                try:
                    gg, _ = grp.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_DEVICE, seed=seed, shard_block=blk, **knobs)
                    ok = np.array_equal(gg.view(np.uint32), ofb.view(np.uint32))
                finally:
                    grp.close()
            line = f"case {case:3d}: tris {n_tri:6d} tex {tex:3d} lights {kw['n_lights']:2d} {W:3d}x{H:3d}x{SPP:2d} depth {depth:2d} {knobs} -> parity {'OK' if ok else 'DIFFERS'}"
            if not ok:
                bad = int((gfb.view(np.uint32) != ofb.view(np.uint32)).any(axis=2).sum())
                print(line, f"({bad} pixels; counters {[k for k in COUNTERS if gst[k] != ost[k]]}); scene kwargs {kw}, seed {seed}", flush=True)
                sys.exit(1)
            # production builds against the oracle
            lo, hi = np.array([-20.0, 0.0, -10.0]), np.array([20.0, 16.0, 10.0])
            rays = random_rays(rng, lo, hi, 4000)
            op, ob = orc.cast_rays(rays)
            prod = []
            for what, build, mode in (("gbest", {}, rt.RT_CAST_EXTEND_GLOBAL), ("lbvh", dict(device_bvh=True), rt.RT_CAST_EXTEND), ("wide-host", dict(wide=True), rt.RT_CAST_EXTEND),
                                      ("wide-dev", dict(wide=True, device_bvh=True), rt.RT_CAST_EXTEND)):
                if sc.n_triangles == 0:  # an empty scene (open room, no random triangles): nothing to build a production tree from
                    break
                sc2 = dev if not build else rt.DeviceScene(sc, **build)
                try:
                    gp, gb, _ = sc2.cast_rays_ex(rays, mode)
                    hit_same = np.array_equal(op == 0xFFFFFFFF, gp == 0xFFFFFFFF)
                    t_bad = int((ob[:, 2].view(np.uint32) != gb[:, 2].view(np.uint32)).sum())
                    closer = int((gb[:, 2] < ob[:, 2]).sum())
                    ties = int(((op != gp) & (ob[:, 2].view(np.uint32) == gb[:, 2].view(np.uint32))).sum())
                    if big:
                        pfb = np.zeros((H, W, 3), dtype=np.float32)
                        sc2.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_DEVICE, seed=seed, global_best=mode == rt.RT_CAST_EXTEND_GLOBAL, shard_index=3, shard_count=16, shard_block=256, out=pfb)
                    else:
                        pfb, _ = sc2.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_DEVICE, seed=seed, global_best=mode == rt.RT_CAST_EXTEND_GLOBAL)
                    rel = np.abs(pfb - ofb) / np.maximum(np.abs(ofb), 1e-3)
                    beyond = int((rel > 1e-5).any(axis=2).sum())
                    worst_prod = max(worst_prod, beyond)
                    farther = t_bad - closer
                    prod.append(f"{what}: t!= {t_bad} (closer {closer}) ties {ties} px>1e-5 {beyond}/{W * H}")
                    if only >= 0:
                        for i in np.nonzero(ob[:, 2].view(np.uint32) != gb[:, 2].view(np.uint32))[0][:8]:
                            print(f"    ray {i}: o {rays[i, :3].tolist()} d {rays[i, 3:].tolist()} oracle prim {op[i]} bct {ob[i].tolist()} | {what} prim {gp[i]} bct {gb[i].tolist()}")
                    dt = np.abs(ob[:, 2].view(np.int32).astype(np.int64) - gb[:, 2].view(np.int32).astype(np.int64))
                    both = (op != 0xFFFFFFFF) & (gp != 0xFFFFFFFF)
                    max_ulp = int(dt[both].max(initial=0))
                    # the production contract: hit / miss as the oracle; a differing t is rounding-sized; the wide tree is never farther
                    broken = (not hit_same) or max_ulp > 4 or (farther > 0 and what.startswith("wide"))
                    if farther > 0:
                        prod[-1] += f" [{farther} farther, max {max_ulp} ulp]"
                    if broken and only < 0:
                        print(line, "| PRODUCTION", what, f"hit/miss same {hit_same}, {farther} farther, max {max_ulp} ulp; scene kwargs {kw}", flush=True)
                        sys.exit(2)
                finally:
                    if build:
                        sc2.close()
            print(line, "|", "; ".join(prod), flush=True)
        finally:
            dev.close()
            orc.close()
    print(f"{n_cases} cases, parity mode bit-exact in all; production images: at most {worst_prod} pixels beyond 1e-5 relative in a case; {time.time() - t_start:.0f} s")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/flake_probe.py — development aid: repeat full-size renders in one process and report any run-to-run difference."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracing-course-hw-public_amd")
sc = rt.scenegen.room_scene(262144, seed=0x5EED5EED, tex_size=256, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0,
                            alpha_fraction=0.02, offset=0.15, camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
dev = rt.DeviceScene(sc)
W = H = 1000
ref = {}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for name, kw, env in (("wf8", dict(samples=8, seed=7), {}), ("wf1c", dict(samples=1, seed=0x5EED5EED, counters=True), {}),
                          ("wf8_nosort_3M", dict(samples=8, seed=7), {"RT_WF_SORT": "0", "RT_WF_MAX_PATHS": "3000000"}),
                          ("shards4", dict(samples=8, seed=7), {"shards": "4"}), ("mega8", dict(samples=8, seed=7, megakernel=True), {})):
        for k, v in env.items():
            if k != "shards":
                os.environ[k] = v
        if "shards" in env:
            fb = np.zeros((H, W, 3), dtype=np.float32)
            for r in range(4):
                dev.run_raytracer(W, H, kw["samples"], seed=kw["seed"], shard_index=r, shard_count=4, shard_block=8 * W, out=fb)
            st = {}
        else:
            fb, st = dev.run_raytracer(W, H, kw.pop("samples"), **kw)
        for k in env:
            os.environ.pop(k, None)
        key = "wf8" if name in ("wf8_nosort_3M", "shards4", "mega8") else name
        if key not in ref:
            ref[key] = (fb.copy(), dict(st))
        else:
            d = (fb.view(np.uint32) != ref[key][0].view(np.uint32)).any(axis=2)
            cnt = {k: (st[k], ref[key][1][k]) for k in st if name == key and k in ("casts", "nodes_visited", "tri_tests", "shaded_hits") and st[k] != ref[key][1][k]}
            if d.any() or cnt:
                ys, xs = np.nonzero(d)
                print(f"rep {rep} {name}: {int(d.sum())} pixels differ, first {list(zip(xs[:5].tolist(), ys[:5].tolist()))} counters {cnt}", flush=True)
    print("rep", rep, "done", flush=True)

#!/usr/bin/env python3
"""tools/dual_probe.py — development aid: do two concurrent half-renders (two scenes = two streams + workspaces, image
shards 0/1 of 2) beat one full render? Tests whether overlapping one stream's memory-bound shade/sort with the other's
VALU-bound extend pays on this GPU."""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracing-course-hw-public_amd")
sc = rt.scenegen.room_scene(262144, seed=0x5EED5EED, tex_size=1024, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0,
                            alpha_fraction=0.02, offset=0.15, camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
W = H = 1000; SPP = 64
devs = [rt.DeviceScene(sc), rt.DeviceScene(sc)]
def full():
    t = time.perf_counter(); fb, _ = devs[0].run_raytracer(W, H, SPP, seed=1); return time.perf_counter() - t, fb
def dual(stagger):
    fb = np.zeros((H, W, 3), dtype=np.float32)
    def work(i):
        if i == 1 and stagger: time.sleep(stagger)
        devs[i].run_raytracer(W, H, SPP, seed=1, shard_index=i, shard_count=2, shard_block=8 * W, out=fb)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; return time.perf_counter() - t, fb
full(); dual(0)
for rep in range(2):
    tf, a = full()
    for st in (0.0, 0.010, 0.020):
        td, b = dual(st)
        print(f"full {tf*1e3:.1f} ms   dual(stagger {st*1e3:.0f} ms) {td*1e3:.1f} ms   same image {bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))}", flush=True)

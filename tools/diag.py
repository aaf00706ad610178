#!/usr/bin/env python3
"""tools_diag.py — development aid: run the -DRT_DIAG variant on the bench scene and print the wave-level census."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "raytracing-course-hw-public_amd/csrc/variants/diag.so")
rt = importlib.import_module("raytracing-course-hw-public_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = rt.scenegen.room_scene(262144, seed=0x5EED5EED, tex_size=1024, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0,
                            alpha_fraction=0.02, offset=0.15, camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
dev = rt.DeviceScene(sc)
dev.run_raytracer(1000, 1000, spp, seed=1, counters=True)
_, st = dev.run_raytracer(1000, 1000, spp, seed=1, counters=True)
n = 1e6 * spp
names = ["trav_iters", "trav_lanes", "inner_execs", "inner_lanes", "tri_execs", "tri_lanes", "exact_execs", "pop_iters", "pop_lanes", "shade_execs", "shade_lanes", "outer_iters"]
keys = ["samples", "casts", "nodes_visited", "box_tests", "tri_tests", "shaded_hits", "light_queries", "light_nodes", "light_box_tests", "light_tri_tests", "light_hits", "texel_fetches"]
v = {a: st[k] for a, k in zip(names, keys)}
print("kernel_ms", st["kernel_ms"])
for a in names: print(f"{a:12s} {v[a]:.4g}  per-sample {v[a]/n:.3f}")
print("lanes/trav_iter", v["trav_lanes"]/v["trav_iters"], " inner lanes/exec", v["inner_lanes"]/max(1,v["inner_execs"]), " tri lanes/exec", v["tri_lanes"]/max(1,v["tri_execs"]),
      " pop lanes/iter", v["pop_lanes"]/max(1,v["pop_iters"]), " shade lanes/exec", v["shade_lanes"]/max(1,v["shade_execs"]))
print("inner_execs/trav_iter", v["inner_execs"]/v["trav_iters"], "tri_execs/trav_iter", v["tri_execs"]/v["trav_iters"], "pop_iters/trav_iter", v["pop_iters"]/v["trav_iters"], "exact/trav_iter", v["exact_execs"]/v["trav_iters"])

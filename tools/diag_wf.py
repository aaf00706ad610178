#!/usr/bin/env python3
"""tools_diag_wf.py — development aid: wave-level census of wf_extend (-DRT_DIAG variant) on the bench scene."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "raytracing-course-hw-public_amd/csrc/variants/%s.so" % os.environ.get("RT_DIAG_VARIANT", "diag"))
rt = importlib.import_module("raytracing-course-hw-public_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = rt.scenegen.room_scene(262144, seed=0x5EED5EED, tex_size=1024, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0,
                            alpha_fraction=0.02, offset=0.15, camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
dev = rt.DeviceScene(sc)
dev.run_raytracer(1000, 1000, spp, seed=1, counters=not os.environ.get("RT_DIAG_NOCOUNTERS"))
out = np.zeros(32, dtype=np.uint64)
lib = rt.lib(); lib.rt_debug_census.argtypes = [C.c_void_p, C.c_void_p]; lib.rt_debug_census(dev._h, out.ctypes.data)
_, st = dev.run_raytracer(1000, 1000, spp, seed=1, counters=not os.environ.get("RT_DIAG_NOCOUNTERS"))
lib.rt_debug_census(dev._h, out.ctypes.data)
n = 1e6 * spp
names = {12: "iterations", 13: "batches", 14: "batch_leaf_lanes", 15: "rounds", 16: "pairs", 17: "hit_path_execs", 18: "node_step_execs", 19: "node_step_lanes", 7: "pop_iters", 8: "pop_lanes", 2: "inner_execs", 3: "inner_lanes", 4: "seqtri_execs", 20: "cyc_refill", 21: "cyc_node_step", 22: "cyc_leaf_batch", 23: "cyc_unwind", 24: "cyc_hit_store", 25: "cyc_wave_total", 26: "waves", 27: "pop_lanes_sum", 28: "stack_overflow_pushes"}
for k, nm in sorted(names.items()):
    print(f"{nm:18s} {float(out[k]):.4g}  per-sample {float(out[k]) / n:.3f}")
print("kernel_ms", st["kernel_ms"], "leaf lanes/batch", float(out[14]) / max(1, float(out[13])), "rounds/batch", float(out[15]) / max(1, float(out[13])),
      "pairs/round", float(out[16]) / max(1, float(out[15])), "lanes/node step", float(out[19]) / max(1, float(out[18])))

tot = float(out[25])
if tot:
    print("wave-cycle shares of wf_extend (s_memtime stamps, all launches): " + ", ".join(f"{nm} {float(out[k]) / tot * 100:.1f} %" for k, nm in ((20, "refill+loop head"), (21, "node step"), (22, "leaf batch"), (23, "unwind"), (24, "hit store"))))
    print("lanes per unwind iteration", float(out[27]) / max(1.0, float(out[7])), "overflow (scratch) pushes per cast", float(out[28]) / max(1, st["casts"] or 1))

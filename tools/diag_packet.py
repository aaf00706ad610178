#!/usr/bin/env python3
"""tools/diag_packet.py [sponza|s10m] [spp] — development aid: trip census of wf_extend_packet (-DRT_DIAG variant 'diag')."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "raytracing-course-hw-public_amd/csrc/variants/diag.so")
os.environ["RT_WF_PACKET"] = "1"
import bench
rt = importlib.import_module("raytracing-course-hw-public_amd")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "sponza"]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = wl["width"], wl["height"]
sc = rt.scenegen.room_scene(wl["triangles"], seed=bench.SEED, tex_size=256, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0, alpha_fraction=0.02,
                            offset=wl["offset"], camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9, aspect=W / H))
dev = rt.DeviceScene(sc)
out = np.zeros(32, dtype=np.uint64)
lib = rt.lib(); lib.rt_debug_census.argtypes = [C.c_void_p, C.c_void_p]
lib.rt_debug_census(dev._h, out.ctypes.data)
_, st = dev.run_raytracer(W, H, spp, seed=1, counters=True)
lib.rt_debug_census(dev._h, out.ctypes.data)
packets = W * H * spp / 64.0
print(f"{wl['label']} {W}x{H}x{spp}: packets {packets:.0f}, trips per packet {float(out[9]) / packets:.1f} (leaf trips {float(out[11]) / packets:.1f}), "
      f"lanes served per trip {float(out[10]) / max(1.0, float(out[9])):.1f}; per-ray: nodes {st['nodes_visited'] / st['casts']:.1f} tri tests {st['tri_tests'] / st['casts']:.1f} (all bounces)")

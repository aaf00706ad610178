#!/usr/bin/env python3
"""tools/diag_shade.py [spp] — development aid: section census of wf_shade (-DRT_DIAG_SHADE variant 'sdiag') on the bench scene."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "raytracing-course-hw-public_amd/csrc/variants/%s.so" % os.environ.get("RT_DIAG_VARIANT", "sdiag"))
rt = importlib.import_module("raytracing-course-hw-public_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = rt.scenegen.room_scene(262144, seed=0x5EED5EED, tex_size=1024, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0,
                            alpha_fraction=0.02, offset=0.15, camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
dev = rt.DeviceScene(sc)
out = np.zeros(32, dtype=np.uint64)
lib = rt.lib(); lib.rt_debug_census.argtypes = [C.c_void_p, C.c_void_p]
dev.run_raytracer(1000, 1000, spp, seed=1, counters=False)
lib.rt_debug_census(dev._h, out.ctypes.data)
_, st = dev.run_raytracer(1000, 1000, spp, seed=1, counters=False)
lib.rt_debug_census(dev._h, out.ctypes.data)
names = ["load order/path/hit", "attrs + material", "4 texture samples", "alpha + direction sample", "vndf pdf + light-BVH pdf", "brdf + early exits", "terminal fold", "queue store"]
tot = float(out[:8].sum())
print(f"wf_shade section census, S-sponza 1000x1000x{spp}: kernel_ms {st['kernel_ms']:.2f}; share of wave cycles, lanes per stamp, stamps")
for i, nm in enumerate(names):
    print(f"  {nm:28s} {float(out[i]) / tot * 100:5.1f} %   {float(out[i]) / max(1.0, float(out[16 + i])):8.0f} cycles/stamp   lanes {float(out[8 + i]) / max(1.0, float(out[16 + i])):5.1f}   stamps {int(out[16 + i])}")
print(f"  total {tot / max(1.0, float(out[16])):.0f} wave cycles per wave-iteration")

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
names = ["load order/path/hit", "attrs + material", "4 texture samples", "alpha coin", "sample: VNDF", "sample: cosine", "sample: light triangle", "vndf pdf",
         "light-BVH pdf (per loop trip)", "brdf + early exits", "terminal store", "queue store"]
N = len(names)
tot = float(out[:N].sum())
lane_tot = float(out[N:2 * N].sum())
print(f"wf_shade section census, S-sponza 1000x1000x{spp}: kernel_ms {st['kernel_ms']:.2f}; share of wave cycles and the lanes each section runs at (lane-weighted cycles / cycles)")
for i, nm in enumerate(names):
    c, lc = float(out[i]), float(out[N + i])
    print(f"  {nm:32s} {c / tot * 100:5.1f} % of wave cycles at {lc / max(1.0, c):5.1f} lanes   -> {max(0.0, c - lc / 64.0) / tot * 100:5.1f} % of the kernel is idle lanes here")
print(f"  whole kernel: {lane_tot / tot:.1f} of 64 lanes on average")

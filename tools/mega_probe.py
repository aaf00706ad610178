import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
rt = importlib.import_module("raytracing-course-hw-public_amd")
sc = rt.scenegen.room_scene(262144, seed=0x5EED5EED, tex_size=1024, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0, alpha_fraction=0.02, offset=0.15, camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
dev = rt.DeviceScene(sc)
for mode, kw in (("device rng megakernel", dict(megakernel=True)), ("reference rng", dict(rng_mode=rt.RT_RNG_REFERENCE))):
    dev.run_raytracer(1000, 1000, 8, seed=1, **kw)
    _, st = dev.run_raytracer(1000, 1000, 16, seed=1, **kw)
    print(mode, "kernel_ms", round(st["kernel_ms"], 1), "Msamples/s", round(16e6 / st["kernel_ms"] / 1e3, 1), flush=True)

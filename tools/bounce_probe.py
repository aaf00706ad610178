import importlib, os, sys
sys.path.insert(0, os.getcwd())
rt = importlib.import_module("raytracing-course-hw-public_amd")
prev=0; prevms=0
for d in range(1,9):
    sc = rt.scenegen.room_scene(262144, seed=0x5EED5EED, tex_size=256, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0, alpha_fraction=0.02, offset=0.15, camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
    sc.ray_depth = d
    dev = rt.DeviceScene(sc)
    dev.run_raytracer(1000,1000,64,seed=1)
    _, t = dev.run_raytracer(1000,1000,64,seed=1)
    _, st = dev.run_raytracer(1000,1000,64,seed=1,counters=True)
    c = st["casts"]/64e6
    print(f"depth {d}: casts/sample {c:.3f} (+{c-prev:.3f})  dominant_ms {t['dominant_ms']:.2f} (+{t['dominant_ms']-prevms:.2f})  nodes/cast {st['nodes_visited']/st['casts']:.1f}", flush=True)
    prev=c; prevms=t['dominant_ms']
    dev.close()

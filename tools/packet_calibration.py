import importlib, os, sys, time
sys.path.insert(0, "/root/repo") if os.path.isdir("/root/repo") else None
sys.path.insert(0, os.getcwd())
import torch
import bench
rt = importlib.import_module("raytracing-course-hw-public_amd")
for wl_name in ("sponza", "s10m"):
    wl = bench.WORKLOADS[wl_name]
    W, H = wl["width"], wl["height"]
    sc = bench.make_scene(rt, wl, wl["triangles"], 64, 1.0)
    dev = rt.DeviceScene(sc, wide=True, device_bvh=(wl_name == "s10m"))
    img = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    for spp in ((4, 8, 16, 32, 64) if wl_name == "sponza" else (8, 16, 32)):
        res = {}
        for pk in ("0", "1"):
            os.environ["RT_WF_PACKET"] = pk
            for _ in range(2):
                dev.run_raytracer_rgb8(W, H, spp, seed=1, device_rgb8=img.data_ptr())
            t0 = time.perf_counter()
            for _ in range(3):
                _, st = dev.run_raytracer_rgb8(W, H, spp, seed=1, device_rgb8=img.data_ptr())
            res[pk] = ((time.perf_counter() - t0) / 3 * 1e3, st["reserved"] / 100.0)
        print(f"{wl_name} wide, {spp} SPP: per-lane {res['0'][0]:.2f} ms, packets {res['1'][0]:.2f} ms ({res['1'][1]:.1f} lanes per trip) -> {res['0'][0] / res['1'][0]:.3f}x", flush=True)
    dev.close()

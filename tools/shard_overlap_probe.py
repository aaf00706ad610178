"""Does overlapping two independent sub-renders on two HIP streams hide each closest-hit launch's drain tail? Two DeviceScene handles
of the same scene (each has its own stream and workspace) render two different 1/C shards from two host threads (ctypes releases the
GIL), against the same two shards rendered one after the other. No library change: the experiment that decides whether a two-stream
pass pipeline is worth building (profiles/r03_variants.txt item 9)."""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rt = importlib.import_module("raytracing-course-hw-public_amd")
wide = "--wide" in sys.argv
W = H = 1000; SPP = 64
sc = rt.scenegen.room_scene(262144, seed=0x5EED5EED, tex_size=1024, n_tex_sets=16, n_materials=64, n_lights=16, light_strength=20.0, alpha_fraction=0.02, offset=0.15,
                            camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9))
devs = [rt.DeviceScene(sc, wide=wide) for _ in range(2)]
imgs = [torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda") for _ in range(2)]
torch.cuda.synchronize()
def one(i, count, reps):
    for _ in range(reps):
        devs[i].run_raytracer_rgb8(W, H, SPP, seed=1, shard_index=i, shard_count=count, shard_block=8 * W, device_rgb8=imgs[i].data_ptr())
def timed(fn, reps=5):
    fn(2)
    t0 = time.perf_counter(); fn(reps)
    return (time.perf_counter() - t0) / reps * 1e3
def serial(count):
    return lambda reps: (one(0, count, reps), one(1, count, reps))
def concurrent(count):
    def f(reps):
        th = [threading.Thread(target=one, args=(i, count, reps)) for i in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
    return f
full = timed(lambda reps: one(0, 1, reps))
for c in (2, 4, 8, 16):
    s, k = timed(serial(c)), timed(concurrent(c))
    print(f"{'wide' if wide else 'parity'}: full {full:.2f} ms; two 1/{c} shards: serial {s:.2f} ms = {s / (2 * full / c):.3f} x ideal, concurrent {k:.2f} ms = {k / (2 * full / c):.3f} x ideal", flush=True)

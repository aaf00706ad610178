"""Development aid (needs a -DRT_WIDE_DIAG variant, RT_AMD_LIB): how the wide traversal's node visits break down on the bench scene."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
rt = importlib.import_module("raytracing-course-hw-public_amd")
from conftest import random_rays
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
sc = rt.scenegen.room_scene(n, seed=0x5EED5EED, tex_size=0, n_materials=64, n_lights=16, offset=0.15 if n < 1000000 else 0.03)
for label, kw in (("host", dict(wide=True)), ("lbvh", dict(wide=True, device_bvh=True))):
    dev = rt.DeviceScene(sc, **kw)
    rays = random_rays(sc, 400000, seed=1)
    _, _, st = dev.cast_rays_ex(rays, rt.RT_CAST_EXTEND)
    v = st["nodes_visited"]
    print(f"{label}: visits/cast {v / len(rays):.1f}  tri tests/cast {st['tri_tests'] / len(rays):.1f}  empty visits {st['light_hits'] / v:.3f} (distance-only {st['light_queries'] / v:.3f})  "
          f"inner children hit per visit {st['light_box_tests'] / v:.2f}  visits with a leaf slot hit {st['light_nodes'] / v:.3f}  kernel {st['kernel_ms']:.2f} ms")
    dev.close()

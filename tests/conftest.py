"""pytest configuration: markers, import paths, shared scene fixtures.

`-m "not gpu"` tests: the oracle against golden vectors, host logic (loader, film, BVH builder through the oracle),
ABI symbol checks. `-m gpu` tests: parity of the HIP path (through the C-ABI) against the oracle on the same
seeded inputs. /root/reference is never read at test time; tests that need the compiled reference
(oracle/_ref/*) skip where it is absent.
"""
import importlib
import os
import sys

import numpy as np
import pytest

try:  # torch (used by a few tests for device buffers / torch.distributed) bundles its own HIP runtime: it must be loaded BEFORE
    import torch  # noqa: F401  librt_amd.so, or the process ends up with two runtimes and torch sees no GPU (same rule as bench.py)
except ImportError:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    return importlib.import_module("raytracing-course-hw-public_amd")


@pytest.fixture(scope="session")
def sg(rt):
    return rt.scenegen


@pytest.fixture(scope="session")
def oracle():
    import oracle as o

    o.lib()
    return o


@pytest.fixture(scope="session")
def gpu(rt):
    """Fails (not skips) when the HIP library or the GPU is missing: there is no CPU fallback to hide behind."""
    rt.lib()
    n = rt.device_count()
    assert n >= 1, "no HIP device visible: -m gpu tests need an MI355X"
    return rt


def golden_scene_specs():
    """The small scenes used for golden vectors and GPU parity (name -> generator kwargs)."""
    return {
        # untextured diffuse + emissive, closed room, alpha pass-through materials
        "room_plain": dict(kind="room", n_random=600, seed=11, n_lights=4, n_materials=8, tex_size=0, alpha_fraction=0.25),
        # textured (colour + normal + MR), smooth normals
        "room_textured": dict(kind="room", n_random=500, seed=12, n_lights=3, n_materials=6, tex_size=16, n_tex_sets=2, alpha_fraction=0.15, smooth_normals=True),
        # no lights at all: cosine-only sampling against the white environment, open scene
        "open_nolight": dict(kind="room", n_random=400, seed=13, n_lights=0, n_materials=6, tex_size=0, open_room=True),
        # S-small: boxes + emissive triangles (axis-aligned, flat AABBs: the hard case for bit-exact traversal)
        "boxes": dict(kind="boxes", n_boxes=24, seed=14, n_lights=3),
        # 40 emissive triangles: a light BVH with inner nodes (19 nodes), so BVH::foreach_intersection's inner-node path
        # (bvh.h:237-260) and uniform_int(0, n - 1) with n = 40 (raytracer.h:353-375) are pinned at more than the
        # bench scene's light count; textured, alpha pass-through materials
        "room_manylights": dict(kind="room", n_random=450, seed=15, n_lights=40, n_materials=6, tex_size=16, n_tex_sets=2, alpha_fraction=0.2, light_strength=6.0),
    }


def make_scene(sg, spec):
    spec = dict(spec)
    kind = spec.pop("kind")
    if kind == "room":
        return sg.room_scene(**spec)
    if kind == "boxes":
        return sg.boxes_scene(**spec)
    raise ValueError(kind)


@pytest.fixture(scope="session")
def scenes(sg):
    return {name: make_scene(sg, spec) for name, spec in golden_scene_specs().items()}


def random_rays(scene, n, seed):
    """Rays with origins inside the scene bounds and random directions, plus some axis-parallel ones."""
    rng = np.random.default_rng(seed)
    lo = scene.positions.reshape(-1, 3).min(axis=0)
    hi = scene.positions.reshape(-1, 3).max(axis=0)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    k = max(1, n // 16)
    axes = np.eye(3, dtype=np.float32)[rng.integers(0, 3, size=k)] * rng.choice([-1.0, 1.0], size=(k, 1)).astype(np.float32)
    d[:k] = axes
    return np.concatenate([o, d.astype(np.float32)], axis=1).astype(np.float32)


SCENE000 = os.path.join(ROOT, "tests", "golden", "txt", "scene-000.txt")  # BASELINE config 1's scene file (a data fixture: the content of the
                                                                          # reference's sample_data/scene-000.txt)


PRACTICE3_5 = os.path.join(ROOT, "tests", "golden", "txt", "practice3_5.txt")  # BASELINE config 2's kind of scene file (a data fixture: the content of the
                                                                                # reference's sample_data/homebrew_primitives/practice3_5.txt: 512x512, 64 SPP,
                                                                                # a Cornell box of 5 PLANEs, 2 BOXes (one emissive) and an ELLIPSOID)


def scene000_box_gltf(rt, sg, out_dir, yfov=1.2):
    """The triangles of scene-000.txt's BOX (the part of config 1 the reference at HEAD can still render: it has no ELLIPSOID / PLANE and no
    scene-txt parser) exported as glTF for the reference binary: same 12 triangles and material, camera at the file's position looking down -z,
    the reference's white environment. Returns (gltf path, arrays of the txt-loaded scene)."""
    a = rt.parse_scene_txt(SCENE000).arrays()
    b = dict(a, primitives=[], bg_color=np.ones(3, dtype=np.float32))
    sc = sg.scene_from_arrays(b, yfov=yfov, rotation=(0.0, 0.0, 0.0, 1.0), face_normals=True)
    return sg.write_gltf(sc, os.path.join(str(out_dir), "scene000_box.gltf")), a


def explain_differing_pixels(gpu, orc, parity_dev, prod_dev, W, H, spp, seed, pixels):
    """Why a production image differs from the parity image in `pixels` (row, col): the oracle replays each pixel's paths (device-RNG mode) and
    logs every ray they cast; the rays go through both scenes' own closest-hit kernels. Up to the first differing hit the production path IS the
    parity path, so that hit names the cause. Allowed by the production contract: an exact tie (t bit-equal, another triangle) or a
    hit the reference's near-local pruning skips (bvh.h:216-223), which the production traversal finds CLOSER by rounding (<= 1e-6 relative).
    Returns one record per pixel; raises on anything else."""
    out = []
    for (y, x) in pixels:
        rays, smp = orc.trace_pixel(W, H, spp, int(y) * W + int(x), seed=seed)
        pp, pb, _ = parity_dev.cast_rays_ex(rays, gpu.RT_CAST_EXTEND)
        qp, qb, _ = prod_dev.cast_rays_ex(rays, gpu.RT_CAST_EXTEND)
        op, ob = orc.cast_rays(rays)
        assert np.array_equal(pp, op) and np.array_equal(pb.view(np.uint32), ob.view(np.uint32))  # the parity scene is the oracle's, also on these rays
        diff = np.nonzero((pp != qp) | (pb[:, 2].view(np.uint32) != qb[:, 2].view(np.uint32)))[0]
        assert len(diff) > 0, f"pixel ({y}, {x}): images differ but every ray of its parity paths has the same closest hit in both scenes"
        k = int(diff[0])
        tp, tq = float(pb[k, 2]), float(qb[k, 2])
        if tp == tq and pp[k] != qp[k]:
            cause = "exact tie: same t, another triangle"
        elif qp[k] != 0xFFFFFFFF and (pp[k] == 0xFFFFFFFF or tq < tp) and (pp[k] == 0xFFFFFFFF or (tp - tq) <= 1e-6 * tp):
            cause = "closer hit the reference's near-local pruning skips (bvh.h:216-223)"
        else:
            raise AssertionError(f"pixel ({y}, {x}), sample {int(smp[k])}, ray {k}: parity hit ({int(pp[k])}, t={tp!r}) vs production hit ({int(qp[k])}, t={tq!r}): outside the production contract")
        out.append({"pixel": (int(y), int(x)), "sample": int(smp[k]), "ray_of_pixel": k, "parity": (int(pp[k]), tp), "production": (int(qp[k]), tq), "cause": cause})
    return out

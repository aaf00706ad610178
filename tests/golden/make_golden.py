#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference (container only).

The reference ships no tests, golden images or glTF assets, so every vector is manufactured here:
scenes come from the deterministic generator (raytracing-course-hw-public_amd/scenegen.py, specs in
tests/conftest.py::golden_scene_specs), are written as glTF, and are fed to
  * oracle/_ref/raytracer_ref  — /root/reference/src/main.cpp compiled as is  -> <name>.ppm
  * oracle/_ref/ref_probe      — harness including the reference headers       -> BVH dump, closest hits, light pdf,
                                                                                  flattened scene.objects
  * oracle/_ref/stdrand_probe  — libstdc++ <random> known answers               -> rng_kat.npz
Only data (inputs / expected outputs) is stored; no reference source text. Run:  python tests/golden/make_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import importlib  # noqa: E402

import oracle  # noqa: E402
from conftest import golden_scene_specs, make_scene, random_rays  # noqa: E402

rt = importlib.import_module("raytracing-course-hw-public_amd")

W, H, SPP = 64, 48, 4
N_RAYS = 4096


def main():
    assert oracle.have_reference_build(), "build oracle/_ref first (make -C oracle)"
    import make_features_gltf

    jobs = [(name, spec) for name, spec in golden_scene_specs().items()] + [("features", None), ("txt_boxes", "txt")]
    for name, spec in jobs:
        with tempfile.TemporaryDirectory() as td:
            if spec == "txt":  # scene-txt fixture with BOX / TRIANGLE primitives only: its triangles, exported as glTF
                a = rt.parse_scene_txt(os.path.join(HERE, "txt", "boxes_only.txt")).arrays()
                sc = rt.scenegen.scene_from_arrays(a, yfov=0.8, rotation=(0.0, 0.0, 0.0, 1.0), face_normals=True)
                gltf = rt.scenegen.write_gltf(sc, os.path.join(td, name + ".gltf"))
            elif spec is None:  # the hand-built loader-feature scene: the glTF itself is a committed fixture
                gltf = make_features_gltf.build(os.path.join(HERE, "features"))
                import types

                sc = types.SimpleNamespace(positions=rt.parse_gltf_scene(gltf, W / H).arrays()["positions"])  # bounds for the random rays
            else:
                sc = make_scene(rt.scenegen, spec)
                gltf = rt.scenegen.write_gltf(sc, os.path.join(td, name + ".gltf"))
            ppm = os.path.join(HERE, f"{name}_{W}x{H}x{SPP}.ppm")
            oracle.run_reference(gltf, W, H, SPP, ppm)
            # BVHs
            oracle.ref_probe("bvh", gltf, W, H, os.path.join(td, "bvh.bin"))
            words = np.fromfile(os.path.join(td, "bvh.bin"), dtype=np.uint32)
            out = {}
            p = 0
            for tag in ("scene", "light"):
                nn, no, root = (int(x) for x in words[p : p + 3])
                p += 3
                out[f"{tag}_root"] = np.uint32(root)
                out[f"{tag}_nodes"] = words[p : p + 10 * nn].reshape(nn, 10).copy()
                p += 10 * nn
                out[f"{tag}_order"] = words[p : p + no].copy()
                p += no
            # primary (pixel-centre) rays and random rays
            oracle.ref_probe("primary", gltf, W, H, os.path.join(td, "prim.bin"))
            pr = np.fromfile(os.path.join(td, "prim.bin"), dtype=np.uint32).reshape(-1, 10)
            out["primary_prim"] = pr[:, 0].copy()
            out["primary_bct"] = pr[:, 1:4].copy().view(np.float32)
            out["primary_rays"] = pr[:, 4:10].copy().view(np.float32)
            rays = random_rays(sc, N_RAYS, seed=2024)
            rays.tofile(os.path.join(td, "rays.bin"))
            oracle.ref_probe("cast", gltf, W, H, os.path.join(td, "rays.bin"), os.path.join(td, "cast.bin"))
            cr = np.fromfile(os.path.join(td, "cast.bin"), dtype=np.uint32).reshape(-1, 10)
            out["cast_rays"] = rays
            out["cast_prim"] = cr[:, 0].copy()
            out["cast_bct"] = cr[:, 1:4].copy().view(np.float32)
            oracle.ref_probe("lightpdf", gltf, W, H, os.path.join(td, "rays.bin"), os.path.join(td, "lp.bin"))
            out["light_pdf"] = np.fromfile(os.path.join(td, "lp.bin"), dtype=np.float32)
            # flattened scene.objects + camera
            oracle.ref_probe("scene", gltf, W, H, os.path.join(td, "scene.bin"))
            sw = np.fromfile(os.path.join(td, "scene.bin"), dtype=np.uint32)
            n = int(sw[0])
            objs = sw[1 : 1 + 33 * n].view(np.float32).reshape(n, 33)
            out["obj_positions"] = objs[:, 0:9].copy()
            out["obj_normals"] = objs[:, 9:18].copy()
            out["obj_texcoords"] = objs[:, 18:24].copy()
            out["obj_tangents"] = objs[:, 24:33].copy()
            out["camera"] = sw[1 + 33 * n : 1 + 33 * n + 13].view(np.float32).copy()
            np.savez_compressed(os.path.join(HERE, f"{name}_probe.npz"), **out)
            print(name, "triangles", n, "nodes", out["scene_nodes"].shape[0], "ppm", os.path.getsize(ppm))

    # libstdc++ <random> known answers
    kat = {}
    for seed in (0, 1, 2, 42, 3906):
        txt = subprocess.check_output([oracle.STDRAND_PROBE, "real", str(seed), "64"]).decode().split()
        kat[f"real_{seed}"] = np.array([float.fromhex(t) for t in txt], dtype=np.float32)
        for bound in (1, 2, 3, 16, 1000):
            txt = subprocess.check_output([oracle.STDRAND_PROBE, "int", str(seed), str(bound), "64"]).decode().split()
            kat[f"int_{seed}_{bound}"] = np.array([int(t) for t in txt], dtype=np.uint32)
    txt = subprocess.check_output([oracle.STDRAND_PROBE, "range", "7", "-1", "1", "64"]).decode().split()
    kat["range_7_m1_1"] = np.array([float.fromhex(t) for t in txt], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "rng_kat.npz"), **kat)
    print("rng_kat ok")


if __name__ == "__main__":
    main()

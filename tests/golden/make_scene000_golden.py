#!/usr/bin/env python3
"""Golden vector for BASELINE config 1 (sample_data/scene-000.txt; container only, needs oracle/_ref).

The reference at HEAD cannot read scene-txt files and has no ELLIPSOID / PLANE, so the only part of config 1 it can
still render is the BOX: its 12 triangles (what csrc/host/txt_loader.cpp makes of "BOX 0.5 0.5 0.5 / POSITION / ROTATION") are exported
as glTF (tests/conftest.py::scene000_box_gltf) and rendered by the UNMODIFIED reference binary -> txt_scene000_box_64x48x4.ppm.
tests/test_scene_txt.py (oracle) and tests/test_gpu_txt.py (HIP path, reference-RNG mode) must reproduce those bytes.
Only data is stored. Run:  python tests/golden/make_scene000_golden.py
"""
import importlib
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle  # noqa: E402
from conftest import scene000_box_gltf  # noqa: E402

rt = importlib.import_module("raytracing-course-hw-public_amd")
W, H, SPP = 64, 48, 4

if __name__ == "__main__":
    assert oracle.have_reference_build(), "build oracle/_ref first (make -C oracle)"
    with tempfile.TemporaryDirectory() as td:
        gltf, a = scene000_box_gltf(rt, rt.scenegen, td)
        ppm = os.path.join(HERE, f"txt_scene000_box_{W}x{H}x{SPP}.ppm")
        img = oracle.run_reference(gltf, W, H, SPP, ppm)
        print("scene-000 box:", a["positions"].shape[0], "triangles ->", ppm, "non-background pixels:", int((img != img[0, 0]).any(axis=2).sum()))

#!/usr/bin/env python3
"""Writes tests/golden/features/features.gltf (+ .bin, PNGs): a hand-built glTF that exercises the loader paths the
generated scenes do not — scene selection, node hierarchy with `matrix` AND TRS on the same node, non-uniform scale,
a camera below transformed parents with its own aspectRatio, triangle strips (mode 5), u8 / u16 / u32 indices, shared
vertices, a primitive without NORMAL / TEXCOORD_0, the lower-case `tangent` attribute the reference reads
(scene.h:336), emissive texture + KHR_materials_emissive_strength, alpha < 1 (-> ior 1.5), a material without
pbrMetallicRoughness, a node outside the selected scene. Input data only; expected outputs come from the reference
(make_golden.py)."""
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

sg = importlib.import_module("raytracing-course-hw-public_amd.scenegen")


def build(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(20240607)
    blob = bytearray()
    views, accessors = [], []

    def add(data: np.ndarray, ctype: int, atype: str, pad_to=4):
        while len(blob) % pad_to:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": data.nbytes})
        blob.extend(data.tobytes())
        accessors.append({"bufferView": len(views) - 1, "componentType": ctype, "count": int(data.shape[0]), "type": atype})
        return len(accessors) - 1

    F, U8, U16, U32 = 5126, 5121, 5123, 5125

    def unit(v):
        v = np.asarray(v, dtype=np.float64)
        return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)

    # mesh 0: a ribbon as a triangle strip (mode 5), u16 indices, NORMAL + TEXCOORD_0 + lower-case tangent
    n_seg = 7
    xs = np.linspace(-2.0, 2.0, n_seg)
    strip_pos = np.array([[x, 0.4 * np.sin(2.0 * x) + (0.0 if k == 0 else 1.1), 0.3 * np.cos(x) + 0.2 * k] for x in xs for k in (0, 1)], dtype=np.float32)
    strip_nrm = unit(np.stack([0.3 * np.cos(2 * strip_pos[:, 0]), np.full(len(strip_pos), 0.2), np.ones(len(strip_pos))], axis=1))
    strip_uv = np.stack([(strip_pos[:, 0] + 2) / 4 * 2.5, strip_pos[:, 1] * 1.7], axis=1).astype(np.float32)
    strip_tan = unit(np.stack([np.ones(len(strip_pos)), 0.8 * np.cos(2 * strip_pos[:, 0]), np.zeros(len(strip_pos))], axis=1))
    strip_idx = np.arange(len(strip_pos), dtype=np.uint16)
    m0 = {"attributes": {"POSITION": add(strip_pos, F, "VEC3"), "NORMAL": add(strip_nrm, F, "VEC3"), "TEXCOORD_0": add(strip_uv, F, "VEC2"),
                         "tangent": add(strip_tan, F, "VEC3")}, "indices": add(strip_idx, U16, "SCALAR"), "material": 0, "mode": 5}
    # mesh 1: a small pyramid, mode 4, u8 indices with shared vertices, no NORMAL / TEXCOORD_0 (flat normals, uv 0)
    pyr_pos = np.array([[-0.5, 0, -0.5], [0.5, 0, -0.5], [0.5, 0, 0.5], [-0.5, 0, 0.5], [0.0, 0.9, 0.0]], dtype=np.float32)
    pyr_idx = np.array([0, 1, 4, 1, 2, 4, 2, 3, 4, 3, 0, 4, 0, 2, 1, 0, 3, 2], dtype=np.uint8)
    m1 = {"attributes": {"POSITION": add(pyr_pos, F, "VEC3")}, "indices": add(pyr_idx, U8, "SCALAR"), "material": 1, "mode": 4}
    # mesh 2: floor + back wall (u32 indices, no "mode" key), an emissive panel (emissive texture + strength), a plain-default material blob
    quad = lambda a, b, c, d: np.array([a, b, c, d], dtype=np.float32)  # noqa: E731
    floor = quad([-6, -1, -6], [6, -1, -6], [6, -1, 6], [-6, -1, 6])
    wall = quad([-6, -1, -6], [-6, 5, -6], [6, 5, -6], [6, -1, -6])
    fw_pos = np.concatenate([floor, wall])
    fw_nrm = np.concatenate([np.tile([0, 1, 0], (4, 1)), np.tile([0, 0, 1], (4, 1))]).astype(np.float32)
    fw_uv = np.tile(np.array([[0, 0], [3, 0], [3, 3], [0, 3]], dtype=np.float32), (2, 1))
    fw_idx = np.array([0, 2, 1, 0, 3, 2, 4, 6, 5, 4, 7, 6], dtype=np.uint32)
    p_floor = {"attributes": {"POSITION": add(fw_pos, F, "VEC3"), "NORMAL": add(fw_nrm, F, "VEC3"), "TEXCOORD_0": add(fw_uv, F, "VEC2")},
               "indices": add(fw_idx, U32, "SCALAR"), "material": 0}
    panel = quad([-1.5, 4.5, -2], [1.5, 4.5, -2], [1.5, 4.5, 1], [-1.5, 4.5, 1])
    panel_uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32)
    p_panel = {"attributes": {"POSITION": add(panel, F, "VEC3"), "TEXCOORD_0": add(panel_uv, F, "VEC2")}, "indices": add(np.array([0, 1, 2, 0, 2, 3], dtype=np.uint16), U16, "SCALAR"),
               "material": 2, "mode": 4}
    blob_pos = (rng.normal(size=(30, 3)) * 0.5 + np.array([2.5, 0.3, 1.0])).astype(np.float32)
    blob_idx = rng.integers(0, 30, size=36).astype(np.uint8)
    p_blob = {"attributes": {"POSITION": add(blob_pos, F, "VEC3")}, "indices": add(blob_idx, U8, "SCALAR"), "material": 3, "mode": 4}

    # textures
    def tex(w, h, kind):
        t = rng.integers(30, 256, size=(h, w, 4), dtype=np.uint8)
        if kind == "normal":
            t[..., 0:2] = rng.integers(96, 160, size=(h, w, 2), dtype=np.uint8)
            t[..., 2] = 255
        t[..., 3] = 255
        return t

    images = []
    for i, (w, h, kind) in enumerate([(8, 8, "color"), (8, 8, "normal"), (8, 8, "mr"), (5, 3, "emissive")]):
        name = f"tex{i}.png"
        sg.write_png_rgba8(os.path.join(out_dir, name), tex(w, h, kind))
        images.append({"uri": name})

    c30, s30 = np.cos(np.pi / 6), np.sin(np.pi / 6)
    # column-major: rotation about y by 30 deg, scale (1.5, 1, 0.75), translation (0.5, 0, -1)
    m_root = [1.5 * c30, 0, -1.5 * s30, 0, 0, 1, 0, 0, 0.75 * s30, 0, 0.75 * c30, 0, 0.5, 0, -1, 1]
    m_child = [1, 0, 0, 0, 0, 0.8, 0.1, 0, 0, -0.1, 0.8, 0, -1.2, 0, 0.8, 1]
    q = lambda ax, ang: [float(a * np.sin(ang / 2)) for a in ax] + [float(np.cos(ang / 2))]  # noqa: E731
    gltf = {
        "asset": {"version": "2.0"},
        "scene": 1,
        "scenes": [{"nodes": [5]}, {"nodes": [0, 4]}],
        "nodes": [
            {"matrix": m_root, "children": [1, 2]},
            {"translation": [0.0, 0.6, 0.5], "rotation": q([0.2672612, 0.5345225, 0.8017837], 0.7), "scale": [0.8, 1.2, 1.0], "mesh": 0},
            {"matrix": m_child, "translation": [0.3, 0.0, 0.4], "rotation": q([0, 1, 0], -0.4), "scale": [1.0, 1.3, 1.0], "mesh": 1, "children": [3]},
            {"camera": 0, "translation": [1.0, 2.2, 7.5], "rotation": q([0.70710678, 0.70710678, 0.0], -0.25)},
            {"mesh": 2},
            {"mesh": 0, "translation": [100, 100, 100]},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.8, "aspectRatio": 1.25, "znear": 0.1}}],
        "meshes": [{"primitives": [m0]}, {"primitives": [m1]}, {"primitives": [p_floor, p_panel, p_blob]}],
        "materials": [
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicRoughnessTexture": {"index": 2}, "roughnessFactor": 0.6, "metallicFactor": 0.3},
             "normalTexture": {"index": 1}},
            {"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.3, 0.2, 0.5], "roughnessFactor": 0.35}},
            {"emissiveFactor": [1.0, 0.9, 0.8], "emissiveTexture": {"index": 3}, "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 12.0}},
             "pbrMetallicRoughness": {"baseColorFactor": [0.1, 0.1, 0.1, 1.0]}},
            {"name": "no_pbr_block"},
        ],
        "textures": [{"source": i} for i in range(4)],
        "images": images,
        "buffers": [{"uri": "features.bin", "byteLength": len(blob)}],
        "bufferViews": views,
        "accessors": accessors,
    }
    with open(os.path.join(out_dir, "features.bin"), "wb") as f:
        f.write(bytes(blob))
    with open(os.path.join(out_dir, "features.gltf"), "w") as f:
        json.dump(gltf, f, indent=1)
    return os.path.join(out_dir, "features.gltf")


if __name__ == "__main__":
    print(build(os.path.join(HERE, "features")))

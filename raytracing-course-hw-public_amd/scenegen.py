"""Deterministic synthetic scenes for the render-loop hot path and a glTF writer.

No glTF asset ships with the reference (sample_data/.gitignore:1 ignores /gltf/), so every scene the parity tests
and bench.py use is generated here from a fixed seed. A scene is a plain `Scene` of numpy arrays in the layout
of include/rt_abi.h; `write_gltf` emits the same triangles as glTF 2.0 + .bin + PNG in exactly the subset the
reference loader reads (src/scene.h:183-501: external buffers, indexed primitives, a material on every
primitive, tightly packed attributes), so the unmodified reference binary can render the very same scene.
"""
from __future__ import annotations

import json
import os
import struct
import zlib
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

SEED = 0x5EED5EED
TEX_NONE = -1


@dataclass
class Material:
    color: tuple = (1.0, 1.0, 1.0, 1.0)
    emission: tuple = (0.0, 0.0, 0.0)  # emissiveFactor (before strength)
    emissive_strength: Optional[float] = None
    roughness: float = 1.0
    metallic: float = 1.0
    ior: float = 1.5
    color_tex: int = TEX_NONE
    emissive_tex: int = TEX_NONE
    metallic_roughness_tex: int = TEX_NONE
    normal_tex: int = TEX_NONE

    def emission_f32(self) -> np.ndarray:
        e = np.asarray(self.emission, dtype=np.float32)
        if self.emissive_strength is not None:  # scene.h:268-273: emission *= float(strength)
            e = e * np.float32(self.emissive_strength)
        return e.astype(np.float32)


@dataclass
class Camera:
    position: np.ndarray
    right: np.ndarray
    up: np.ndarray
    forward: np.ndarray
    fov_x: float
    # glTF side (only used by write_gltf)
    yfov: float = 0.9
    rotation: tuple = (0.0, 0.0, 0.0, 1.0)  # quaternion xyzw


@dataclass
class Scene:
    positions: np.ndarray  # (N,3,3) f32
    normals: Optional[np.ndarray]  # (N,3,3) f32 or None -> geometric normal (scene.h:427-430)
    texcoords: np.ndarray  # (N,3,2) f32
    tangents: np.ndarray  # (N,3,3) f32
    material_ids: np.ndarray  # (N,) u32
    materials: List[Material]
    textures: List[np.ndarray] = field(default_factory=list)  # each (H,W,4) u8
    camera: Optional[Camera] = None
    bg_color: tuple = (1.0, 1.0, 1.0)
    bg_texture: int = -1  # Scene::bg (scene.h:81): index into textures of the environment map, -1 = the 1x1 white default
    ray_depth: int = 8
    # analytic primitives of the scene-txt front end: dicts {kind, material_id, param[3], position[3], rotation xyzw[4]}
    primitives: List[dict] = field(default_factory=list)

    @property
    def n_triangles(self) -> int:
        return int(self.positions.shape[0])

    def resolved_normals(self) -> np.ndarray:
        """Per-vertex normals as the reference loader stores them (scene.h:392-397, 423-430)."""
        if self.normals is not None:
            return _normalize_f32(self.normals.astype(np.float32))
        v = (self.positions[:, 1] - self.positions[:, 0]).astype(np.float32)
        u = (self.positions[:, 2] - self.positions[:, 0]).astype(np.float32)
        n = _normalize_f32(_cross_f32(v, u))
        return np.repeat(n[:, None, :], 3, axis=1).astype(np.float32)


def _cross_f32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    x = a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1]
    y = a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2]
    z = a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
    return np.stack([x, y, z], axis=-1).astype(np.float32)


def _normalize_f32(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.float32)
    l2 = v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1] + v[..., 2] * v[..., 2]
    l = np.sqrt(l2.astype(np.float32)).astype(np.float32)
    return (v / l[..., None]).astype(np.float32)


def look_camera(position, yaw_deg: float = 0.0, yfov: float = 0.9, aspect: float = 1.0) -> Camera:
    """Camera rotated about +y by `yaw_deg` from the glTF default (looking down -z). The basis vectors are
    what the reference derives from the node rotation matrix (geometry.h:179-196, scene.h:243-254) for exact
    half-angle quaternions; for the direct-ABI path (bench) any orthonormal basis is legal input anyway."""
    half = np.float32(np.deg2rad(yaw_deg) / 2.0)
    qy, qw = np.float32(np.sin(half)), np.float32(np.cos(half))
    x, y, z, w = np.float32(0), qy, np.float32(0), qw
    one, two = np.float32(1), np.float32(2)
    m = np.array(
        [
            [one - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w)],
            [two * (x * y + z * w), one - two * (x * x + z * z), two * (y * z - x * w)],
            [two * (x * z - y * w), two * (y * z + x * w), one - two * (x * x + y * y)],
        ],
        dtype=np.float32,
    )
    fwd = _normalize_f32(m @ np.array([0, 0, -1], dtype=np.float32))
    up = _normalize_f32(m @ np.array([0, 1, 0], dtype=np.float32))
    right = _normalize_f32(m @ np.array([1, 0, 0], dtype=np.float32))
    fov_x = float(np.float32(np.arctan(np.tan(np.float32(yfov) / np.float32(2)) * np.float32(aspect)) * np.float32(2)))
    return Camera(
        position=np.asarray(position, dtype=np.float32),
        right=right,
        up=up,
        forward=fwd,
        fov_x=fov_x,
        yfov=yfov,
        rotation=(0.0, float(qy), 0.0, float(qw)),
    )


# ------------------------------------------------------------------------------------------------ textures
def value_noise_texture(rng: np.random.Generator, size: int, kind: str) -> np.ndarray:
    """Procedural RGBA8 texture: smooth value noise. kind in {color, normal, mr}."""
    def noise(channels: int, cells: int) -> np.ndarray:
        g = rng.random((cells, cells, channels), dtype=np.float32)
        g = np.concatenate([g, g[:1]], axis=0)
        g = np.concatenate([g, g[:, :1]], axis=1)
        t = np.linspace(0, cells, size, endpoint=False, dtype=np.float32)
        i = np.floor(t).astype(np.int64)
        f = (t - i).astype(np.float32)
        f = f * f * (3 - 2 * f)
        a = g[i][:, i]
        b = g[i + 1][:, i]
        c = g[i][:, i + 1]
        d = g[i + 1][:, i + 1]
        fy = f[:, None, None]
        fx = f[None, :, None]
        return (a * (1 - fy) * (1 - fx) + b * fy * (1 - fx) + c * (1 - fy) * fx + d * fy * fx).astype(np.float32)

    if kind == "color":
        rgb = 0.15 + 0.8 * (0.6 * noise(3, 8) + 0.4 * noise(3, 32))
        a = np.ones((size, size, 1), dtype=np.float32)
        img = np.concatenate([rgb, a], axis=2)
    elif kind == "normal":
        xy = 0.5 + 0.25 * (noise(2, 16) - 0.5)
        z = np.ones((size, size, 1), dtype=np.float32)
        a = np.ones((size, size, 1), dtype=np.float32)
        img = np.concatenate([xy, z, a], axis=2)
    elif kind == "mr":
        n = noise(2, 16)
        r = np.zeros((size, size, 1), dtype=np.float32)
        g = 0.2 + 0.8 * n[..., :1]  # roughness in G
        b = n[..., 1:2]  # metallic in B
        a = np.ones((size, size, 1), dtype=np.float32)
        img = np.concatenate([r, g, b, a], axis=2)
    else:
        raise ValueError(kind)
    return np.clip(np.round(img * 255.0), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ geometry
def _box_triangles(lo, hi) -> np.ndarray:
    lo = np.asarray(lo, dtype=np.float32)
    hi = np.asarray(hi, dtype=np.float32)
    c = np.array([[lo[0] if not (i & 1) else hi[0], lo[1] if not (i & 2) else hi[1], lo[2] if not (i & 4) else hi[2]] for i in range(8)], dtype=np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = []
    for a, b, cc, d in quads:
        tris.append([c[a], c[b], c[cc]])
        tris.append([c[a], c[cc], c[d]])
    return np.asarray(tris, dtype=np.float32)


def _default_attrs(n: int):
    tex = np.zeros((n, 3, 2), dtype=np.float32)
    tan = np.zeros((n, 3, 3), dtype=np.float32)
    tan[..., 0] = 1.0
    return tex, tan


def room_scene(
    n_random: int,
    seed: int = SEED,
    room=((-20.0, 0.0, -10.0), (20.0, 16.0, 10.0)),
    offset: float = 0.15,
    n_lights: int = 16,
    light_strength: float = 20.0,
    n_materials: int = 64,
    tex_size: int = 0,
    n_tex_sets: int = 16,
    alpha_fraction: float = 0.02,
    smooth_normals: bool = False,
    camera: Optional[Camera] = None,
    open_room: bool = False,
) -> Scene:
    """S-sponza family: a closed box room + `n_random` random triangles with centres uniform in the
    room and vertex offsets uniform in +-offset, `n_lights` emissive triangles under the ceiling, `n_materials`
    metallic-roughness materials; with tex_size > 0, `n_tex_sets` procedural (colour, normal, MR) texture sets of
    tex_size^2 RGBA8 are attached round-robin and texcoords are random."""
    rng = np.random.default_rng(seed)
    lo = np.asarray(room[0], dtype=np.float32)
    hi = np.asarray(room[1], dtype=np.float32)
    materials: List[Material] = []
    textures: List[np.ndarray] = []
    if tex_size > 0:
        for _ in range(n_tex_sets):
            textures.append(value_noise_texture(rng, tex_size, "color"))
            textures.append(value_noise_texture(rng, tex_size, "normal"))
            textures.append(value_noise_texture(rng, tex_size, "mr"))
    # material 0: room walls (diffuse), material 1: lights
    materials.append(Material(color=(0.75, 0.75, 0.75, 1.0), roughness=1.0, metallic=0.0))
    materials.append(Material(color=(1.0, 1.0, 1.0, 1.0), emission=(1.0, 0.95, 0.85), emissive_strength=light_strength, roughness=1.0, metallic=0.0))
    for i in range(n_materials):
        col = rng.uniform(0.2, 0.95, size=3)
        alpha = 0.5 if rng.random() < alpha_fraction else 1.0
        m = Material(
            color=(float(col[0]), float(col[1]), float(col[2]), alpha),
            roughness=float(rng.uniform(0.05, 1.0)),
            metallic=float(rng.choice([0.0, 0.0, 1.0, 0.5])),
        )
        if tex_size > 0:
            s = i % n_tex_sets
            m.color_tex, m.normal_tex, m.metallic_roughness_tex = 3 * s, 3 * s + 1, 3 * s + 2
            m.roughness, m.metallic = 1.0, float(rng.choice([0.0, 1.0]))
        materials.append(m)

    parts = []
    mats = []
    if not open_room:
        walls = _box_triangles(lo, hi)
        parts.append(walls)
        mats.append(np.zeros(len(walls), dtype=np.uint32))
    # lights: triangles just under the ceiling
    if n_lights > 0:
        lc = np.stack(
            [rng.uniform(lo[0] + 2, hi[0] - 2, n_lights), np.full(n_lights, hi[1] - 0.05), rng.uniform(lo[2] + 2, hi[2] - 2, n_lights)], axis=1
        ).astype(np.float32)
        lt = np.stack(
            [lc + np.array([-0.8, 0, -0.6], dtype=np.float32), lc + np.array([0.8, 0, -0.6], dtype=np.float32), lc + np.array([0.0, 0, 0.9], dtype=np.float32)],
            axis=1,
        ).astype(np.float32)
        parts.append(lt)
        mats.append(np.ones(n_lights, dtype=np.uint32))
    if n_random > 0:
        centers = rng.uniform(lo + 0.3, hi - 0.3, size=(n_random, 3)).astype(np.float32)
        offs = rng.uniform(-offset, offset, size=(n_random, 3, 3)).astype(np.float32)
        tri = (centers[:, None, :] + offs).astype(np.float32)
        parts.append(tri)
        mats.append(rng.integers(2, 2 + n_materials, size=n_random).astype(np.uint32))
    if parts:
        positions = np.concatenate(parts, axis=0).astype(np.float32)
        material_ids = np.concatenate(mats, axis=0).astype(np.uint32)
    else:  # the empty scene (bvh.h:373-376)
        positions = np.zeros((0, 3, 3), dtype=np.float32)
        material_ids = np.zeros(0, dtype=np.uint32)
    positions = positions + np.float32(0.0)  # canonicalise -0.0 -> +0.0 (identity node transform does that too)
    n = positions.shape[0]
    tex, tan = _default_attrs(n)
    if tex_size > 0:
        base = rng.uniform(0, 4, size=(n, 1, 2)).astype(np.float32)
        tex = (base + rng.uniform(-0.3, 0.3, size=(n, 3, 2)).astype(np.float32)).astype(np.float32)
    normals = None
    if smooth_normals:
        v = positions[:, 1] - positions[:, 0]
        u = positions[:, 2] - positions[:, 0]
        g = _normalize_f32(_cross_f32(v, u))
        jitter = rng.normal(0, 0.25, size=(n, 3, 3)).astype(np.float32)
        normals = _normalize_f32(g[:, None, :] + jitter)
    if camera is None:
        camera = look_camera((lo[0] + 5.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9)  # at (-15,4,0) looking +x
    return Scene(positions, normals, tex, tan, material_ids, materials, textures, camera)


def boxes_scene(n_boxes: int = 64, seed: int = SEED, n_lights: int = 4) -> Scene:
    """S-small: `n_boxes` axis-aligned boxes (12 triangles each) + emissive triangles, open scene."""
    rng = np.random.default_rng(seed)
    materials = [
        Material(color=(0.8, 0.8, 0.8, 1.0), roughness=1.0, metallic=0.0),
        Material(color=(1, 1, 1, 1), emission=(1, 1, 1), emissive_strength=15.0, roughness=1.0, metallic=0.0),
    ]
    for _ in range(8):
        c = rng.uniform(0.2, 0.9, 3)
        materials.append(Material(color=(float(c[0]), float(c[1]), float(c[2]), 1.0), roughness=float(rng.uniform(0.1, 1)), metallic=float(rng.choice([0.0, 1.0]))))
    parts, mats = [], []
    floor = _box_triangles((-12, -0.5, -12), (12, 0.0, 12))
    parts.append(floor)
    mats.append(np.zeros(12, dtype=np.uint32))
    for _ in range(n_boxes):
        c = rng.uniform((-8, 0.3, -8), (8, 5, 8))
        h = rng.uniform(0.2, 0.7, 3)
        parts.append(_box_triangles(c - h, c + h))
        mats.append(np.full(12, rng.integers(2, len(materials)), dtype=np.uint32))
    for i in range(n_lights):
        c = np.array([rng.uniform(-6, 6), 7.5, rng.uniform(-6, 6)], dtype=np.float32)
        parts.append(np.array([[c + [-1, 0, -1], c + [1, 0, -1], c + [0, 0, 1.2]]], dtype=np.float32))
        mats.append(np.ones(1, dtype=np.uint32))
    positions = (np.concatenate(parts).astype(np.float32)) + np.float32(0.0)
    material_ids = np.concatenate(mats).astype(np.uint32)
    tex, tan = _default_attrs(len(positions))
    cam = look_camera((0.0, 3.0, 14.0), yaw_deg=0.0, yfov=0.8)
    return Scene(positions, None, tex, tan, material_ids, materials, [], cam)


# ------------------------------------------------------------------------------------------------ PNG / glTF
def write_png_rgba8(path: str, img: np.ndarray) -> None:
    h, w, c = img.shape
    assert c == 4 and img.dtype == np.uint8
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


def write_gltf(scene: Scene, path: str) -> str:
    """Write `scene` as <path>.gltf + .bin (+ PNGs) readable by the reference loader. One mesh primitive per
    material, unshared vertices, u32 indices, identity mesh node, camera node with translation + rotation."""
    base = os.path.splitext(path)[0]
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    name = os.path.basename(base)
    bin_chunks: List[bytes] = []
    buffer_views, accessors, primitives = [], [], []
    offset = 0

    def add_view(data: bytes) -> int:
        nonlocal offset
        pad = (-offset) % 4
        if pad:
            bin_chunks.append(b"\x00" * pad)
            offset += pad
        buffer_views.append({"buffer": 0, "byteOffset": offset, "byteLength": len(data)})
        bin_chunks.append(data)
        offset += len(data)
        return len(buffer_views) - 1

    def add_accessor(view: int, ctype: int, count: int, typ: str) -> int:
        accessors.append({"bufferView": view, "componentType": ctype, "count": count, "type": typ})
        return len(accessors) - 1

    order = np.argsort(scene.material_ids, kind="stable")
    # NOTE: object order in the reference = primitive order, so emit primitives as runs of equal material in the
    # ORIGINAL triangle order to keep scene.objects identical to `scene`'s order.
    ids = scene.material_ids
    runs = []
    start = 0
    for i in range(1, len(ids) + 1):
        if i == len(ids) or ids[i] != ids[start]:
            runs.append((start, i))
            start = i
    del order
    for (a, b) in runs:
        n = b - a
        pos = scene.positions[a:b].reshape(n * 3, 3).astype("<f4")
        attrs = {"POSITION": add_accessor(add_view(pos.tobytes()), 5126, n * 3, "VEC3")}
        if scene.normals is not None:
            attrs["NORMAL"] = add_accessor(add_view(scene.normals[a:b].reshape(n * 3, 3).astype("<f4").tobytes()), 5126, n * 3, "VEC3")
        if np.any(scene.texcoords[a:b] != 0):
            attrs["TEXCOORD_0"] = add_accessor(add_view(scene.texcoords[a:b].reshape(n * 3, 2).astype("<f4").tobytes()), 5126, n * 3, "VEC2")
        idx = np.arange(n * 3, dtype="<u4")
        ia = add_accessor(add_view(idx.tobytes()), 5125, n * 3, "SCALAR")
        primitives.append({"attributes": attrs, "indices": ia, "material": int(ids[a]), "mode": 4})

    mats_json = []
    for m in scene.materials:
        pbr = {"baseColorFactor": [float(x) for x in m.color], "roughnessFactor": float(m.roughness), "metallicFactor": float(m.metallic)}
        if m.color_tex != TEX_NONE:
            pbr["baseColorTexture"] = {"index": m.color_tex}
        if m.metallic_roughness_tex != TEX_NONE:
            pbr["metallicRoughnessTexture"] = {"index": m.metallic_roughness_tex}
        mj = {"pbrMetallicRoughness": pbr}
        if any(e != 0 for e in m.emission):
            mj["emissiveFactor"] = [float(x) for x in m.emission]
        if m.emissive_strength is not None:
            mj["extensions"] = {"KHR_materials_emissive_strength": {"emissiveStrength": float(m.emissive_strength)}}
        if m.emissive_tex != TEX_NONE:
            mj["emissiveTexture"] = {"index": m.emissive_tex}
        if m.normal_tex != TEX_NONE:
            mj["normalTexture"] = {"index": m.normal_tex}
        mats_json.append(mj)

    images, textures = [], []
    for i, t in enumerate(scene.textures):
        fn = f"{name}_tex{i}.png"
        write_png_rgba8(os.path.join(d, fn), t)
        images.append({"uri": fn})
        textures.append({"source": i})

    cam = scene.camera
    nodes = [{"mesh": 0}]
    doc = {
        "asset": {"version": "2.0"},
        "scene": 0,
        "scenes": [{"nodes": [0, 1]}],
        "nodes": nodes,
        "meshes": [{"primitives": primitives}],
        "materials": mats_json,
        "buffers": [{"uri": name + ".bin", "byteLength": offset}],
        "bufferViews": buffer_views,
        "accessors": accessors,
        "textures": textures,
        "images": images,
    }
    nodes.append({"camera": 0, "translation": [float(x) for x in cam.position], "rotation": [float(x) for x in cam.rotation]})
    doc["cameras"] = [{"type": "perspective", "perspective": {"yfov": float(cam.yfov), "znear": 0.01}}]
    with open(base + ".bin", "wb") as f:
        f.write(b"".join(bin_chunks))
    with open(base + ".gltf", "w") as f:
        json.dump(doc, f)
    return base + ".gltf"


def scene_from_arrays(a: dict, yfov: float = 0.9, rotation=(0.0, 0.0, 0.0, 1.0), face_normals: bool = False) -> Scene:
    """A Scene from the arrays of a host loader (LoadedScene.arrays()), e.g. to export a scene-txt file's triangles as glTF
    for the reference binary (`yfov` / `rotation` are what write_gltf puts on the camera node: they must describe the same
    camera as a['camera'], which glTF cannot express directly). face_normals=True drops the per-vertex normals so that every
    consumer derives the face normal itself (a glTF primitive without NORMAL, scene.h:427-430)."""
    mats = []
    for m in a["materials"]:
        mats.append(Material(color=tuple(float(x) for x in m["color"]), emission=tuple(float(x) for x in m["emission"]), emissive_strength=None,
                             roughness=float(m["roughness"]), metallic=float(m["metallic"]), ior=float(m["ior"]), color_tex=int(m["color_tex"]),
                             emissive_tex=int(m["emissive_tex"]), metallic_roughness_tex=int(m["metallic_roughness_tex"]), normal_tex=int(m["normal_tex"])))
    c = a["camera"]
    cam = Camera(position=c["position"], right=c["right"], up=c["up"], forward=c["forward"], fov_x=float(c["fov_x"]), yfov=yfov, rotation=rotation)
    return Scene(positions=a["positions"], normals=None if face_normals else a["normals"], texcoords=a["texcoords"], tangents=a["tangents"], material_ids=a["material_ids"], materials=mats,
                 textures=list(a["textures"]), camera=cam, bg_color=tuple(float(x) for x in a["bg_color"]), ray_depth=int(a["ray_depth"]),
                 primitives=[dict(p) for p in a.get("primitives", [])])

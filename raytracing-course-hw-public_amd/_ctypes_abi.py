"""ctypes mirror of include/rt_abi.h and include/rt_host.h (struct layouts + prototypes).

Shared by the product binding (this package) and by the test-side checker binding, which maps the SAME struct
layouts onto its own entry points. Only declarations live here — no compute.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

RT_ABI_VERSION = 4
RT_PRIM_ELLIPSOID = 1
RT_PRIM_PLANE = 2
RT_BUILD_REFERENCE = 0
RT_BUILD_DEVICE_LBVH = 1
RT_BUILD_WIDE = 2
RT_BUILD_WIDE_HOST_COLLAPSE = 4
RT_BUILD_LIGHTS_GLOBAL = 8
RT_BUILD_GROUP_COPY = 16
RT_BUILD_GROUP_SELF_EXCHANGE = 32
RT_BUILDER_PLOC, RT_BUILDER_LBVH = 0, 1
RT_WIDE_ORDER_DEFAULT, RT_WIDE_ORDER_LEVEL, RT_WIDE_ORDER_DFS, RT_WIDE_ORDER_TREELET = range(4)
RT_SORT_AUTO, RT_SORT_OFF, RT_SORT_CELL_OCTANT, RT_SORT_COARSE_CELL_DIR, RT_SORT_OCTANT_CELL, RT_SORT_CELL_OCTANT_CONE, RT_SORT_OCTANT_CELL_CONE, RT_SORT_OCTANT_FINE_CELL_CONE = range(8)
RT_PACKET_AUTO, RT_PACKET_OFF, RT_PACKET_ON = range(3)
RT_TEX_NONE = -1
RT_RNG_DEVICE = 0
RT_RNG_REFERENCE = 1
RT_FLAG_DEVICE_FB = 1
RT_FLAG_COUNTERS = 2
RT_FLAG_MEGAKERNEL = 4
RT_FLAG_GLOBAL_BEST = 8
RT_CAST_PROBE, RT_CAST_EXTEND, RT_CAST_EXTEND_GLOBAL, RT_CAST_PACKET, RT_CAST_PACKET_GLOBAL = range(5)

RT_OK = 0
ERROR_NAMES = {
    0: "RT_OK",
    1: "RT_ERR_INVALID_ARG",
    2: "RT_ERR_NO_DEVICE",
    3: "RT_ERR_HIP",
    4: "RT_ERR_OOM",
    5: "RT_ERR_IO",
    6: "RT_ERR_FORMAT",
    7: "RT_ERR_COMM",
    8: "RT_ERR_UNSUPPORTED",
}

c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)
c_u8_p = C.POINTER(C.c_uint8)


class RtCamera(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("right", C.c_float * 3),
        ("up", C.c_float * 3),
        ("forward", C.c_float * 3),
        ("fov_x", C.c_float),
    ]


class RtTextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba8", c_u8_p)]


class RtMaterialDesc(C.Structure):
    _fields_ = [
        ("color", C.c_float * 4),
        ("emission", C.c_float * 3),
        ("roughness", C.c_float),
        ("metallic", C.c_float),
        ("ior", C.c_float),
        ("color_tex", C.c_int32),
        ("emissive_tex", C.c_int32),
        ("metallic_roughness_tex", C.c_int32),
        ("normal_tex", C.c_int32),
    ]


class RtPrimitiveDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("material_id", C.c_uint32),
        ("param", C.c_float * 3),
        ("position", C.c_float * 3),
        ("rotation", C.c_float * 4),
    ]


class RtBuildOptions(C.Structure):
    _fields_ = [
        ("device_builder", C.c_uint32),
        ("ploc_radius", C.c_uint32),
        ("lbvh_leaf_tris", C.c_uint32),
        ("node_order", C.c_uint32),
        ("wide_cost_node", C.c_float),
        ("wide_cost_tri", C.c_float),
        ("wide_order", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_triangles", C.c_uint32),
        ("positions", c_float_p),
        ("normals", c_float_p),
        ("texcoords", c_float_p),
        ("tangents", c_float_p),
        ("material_ids", c_u32_p),
        ("n_materials", C.c_uint32),
        ("materials", C.POINTER(RtMaterialDesc)),
        ("n_textures", C.c_uint32),
        ("textures", C.POINTER(RtTextureDesc)),
        ("camera", RtCamera),
        ("bg_color", C.c_float * 3),
        ("ray_depth", C.c_uint32),
        ("n_primitives", C.c_uint32),
        ("primitives", C.POINTER(RtPrimitiveDesc)),
        ("build_flags", C.c_uint32),
        ("bg_texture", C.c_int32),
        ("build", RtBuildOptions),
    ]


RT_PROGRESS_FN = C.CFUNCTYPE(None, C.c_uint32, C.c_uint32, C.c_void_p)


class RtParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("samples", C.c_uint32),
        ("rng_mode", C.c_uint32),
        ("seed", C.c_uint64),
        ("shard_index", C.c_uint32),
        ("shard_count", C.c_uint32),
        ("shard_block", C.c_uint32),
        ("flags", C.c_uint32),
        ("sort_mode", C.c_uint32),
        ("packet_mode", C.c_uint32),
        ("packet_min_lanes", C.c_float),
        ("reserved0", C.c_uint32),
        ("max_paths", C.c_uint64),
        ("progress", RT_PROGRESS_FN),
        ("progress_user", C.c_void_p),
    ]


class RtStats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("casts", C.c_uint64),
        ("nodes_visited", C.c_uint64),
        ("box_tests", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("shaded_hits", C.c_uint64),
        ("light_queries", C.c_uint64),
        ("light_nodes", C.c_uint64),
        ("light_box_tests", C.c_uint64),
        ("light_tri_tests", C.c_uint64),
        ("light_hits", C.c_uint64),
        ("texel_fetches", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("total_ms", C.c_double),
        ("dominant_ms", C.c_double),
        ("dominant_launches", C.c_uint32),
        ("packet_lanes_x100", C.c_uint32),
        ("passes", C.c_uint32),
        ("packet_passes", C.c_uint32),
    ]

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


# Every symbol include/rt_abi.h declares (checked by tests/test_abi_symbols.py against the header text).
ABI_PROTOTYPES = {
    "rt_create": (C.c_int, [C.POINTER(RtSceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_destroy": (None, [C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(RtParams), C.c_void_p, C.POINTER(RtStats)]),
    "rt_cast_rays": (C.c_int, [C.c_void_p, c_float_p, C.c_uint32, c_u32_p, c_float_p]),
    "rt_cast_rays_ex": (C.c_int, [C.c_void_p, c_float_p, C.c_uint32, C.c_uint32, c_u32_p, c_float_p, C.POINTER(RtStats)]),
    "rt_surface_normals": (C.c_int, [C.c_void_p, c_float_p, C.c_uint32, c_u32_p, c_float_p, c_float_p, c_float_p]),
    "rt_light_pdf": (C.c_int, [C.c_void_p, c_float_p, C.c_uint32, c_float_p]),
    "rt_bg_at": (C.c_int, [C.c_void_p, c_float_p, C.c_uint32, c_float_p]),
    "rt_bvh_wide_dump": (C.c_int, [C.c_void_p, c_u32_p, c_u32_p, c_u32_p, c_u32_p, c_u32_p]),
    "rt_bvh_info": (C.c_int, [C.c_void_p, C.c_int, c_u32_p, c_u32_p, c_u32_p, c_u32_p, c_u32_p]),
    "rt_tonemap_rgb8": (None, [c_float_p, C.c_size_t, c_u8_p]),
    "rt_render_rgb8": (C.c_int, [C.c_void_p, C.POINTER(RtParams), C.c_void_p, C.POINTER(RtStats)]),
    "rt_film_rgb8": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t, c_u8_p]),
    "rt_last_error": (C.c_char_p, []),
    "rt_source_stamp": (C.c_char_p, []),
    "rt_abi_version": (C.c_uint32, []),
    "rt_device_count": (C.c_int, []),
    "rt_create_on": (C.c_int, [C.POINTER(RtSceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_scene_device_count": (C.c_int, [C.c_void_p]),
    "rt_bvh_device_dump": (C.c_int, [C.c_void_p, C.c_int, c_u32_p, c_u32_p, c_u32_p, c_u32_p, c_u32_p]),
    "rt_build_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rt_build_times_ex": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}
HOST_PROTOTYPES = {
    "rt_gltf_load": (C.c_int, [C.c_char_p, C.c_float, C.POINTER(C.c_void_p)]),
    "rt_txt_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rt_scene_load": (C.c_int, [C.c_char_p, C.c_float, C.POINTER(C.c_void_p)]),
    "rt_loaded_info": (C.c_int, [C.c_void_p, c_u32_p, c_u32_p, c_u32_p, c_u32_p]),
    "rt_loaded_desc": (C.POINTER(RtSceneDesc), [C.c_void_p]),
    "rt_loaded_set_env_map": (C.c_int, [C.c_void_p, C.c_char_p, C.c_float]),
    "rt_loaded_add_light_triangle": (C.c_int, [C.c_void_p, c_float_p, C.c_float]),
    "rt_loaded_disable_textures": (C.c_int, [C.c_void_p]),
    "rt_hdr_decode_file": (C.c_int, [C.c_char_p, c_u32_p, c_u32_p, C.POINTER(c_u8_p)]),
    "rt_loaded_free": (None, [C.c_void_p]),
    "rt_write_ppm": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, c_u8_p]),
    "rt_png_decode_file": (C.c_int, [C.c_char_p, c_u32_p, c_u32_p, C.POINTER(c_u8_p)]),
    "rt_jpeg_decode_file": (C.c_int, [C.c_char_p, c_u32_p, c_u32_p, C.POINTER(c_u8_p)]),
    "rt_image_decode_file": (C.c_int, [C.c_char_p, c_u32_p, c_u32_p, C.POINTER(c_u8_p)]),
    "rt_free": (None, [C.c_void_p]),
    "rt_film_table": (C.c_int, [c_float_p, c_u32_p]),
    "rt_bvh_build_host": (C.c_int, [c_float_p, C.c_uint32, c_u32_p, C.c_uint32, c_u32_p, c_u32_p, c_u32_p, c_u32_p]),
    "rt_bvh_wide_build_host": (C.c_int, [c_float_p, C.c_uint32, C.c_float, C.c_float, c_u32_p, c_u32_p, C.POINTER(C.c_double), c_u32_p, C.c_uint32, c_u32_p]),
}


def bind(lib: C.CDLL, protos: dict, rename=None) -> None:
    for name, (res, args) in protos.items():
        sym = rename(name) if rename else name
        fn = getattr(lib, sym)
        fn.restype = res
        fn.argtypes = args


def as_f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def fptr(a: np.ndarray):
    return a.ctypes.data_as(c_float_p)


def u32ptr(a: np.ndarray):
    return a.ctypes.data_as(c_u32_p)


def u8ptr(a: np.ndarray):
    return a.ctypes.data_as(c_u8_p)


class DescHolder:
    """Builds an rt_scene_desc from a scenegen.Scene-like object and keeps every backing array alive."""

    def __init__(self, scene):
        n = scene.n_triangles
        self.positions = as_f32(scene.positions).reshape(n, 9)
        self.normals = as_f32(scene.resolved_normals()).reshape(n, 9)
        self.texcoords = as_f32(scene.texcoords).reshape(n, 6)
        self.tangents = as_f32(scene.tangents).reshape(n, 9)
        self.material_ids = np.ascontiguousarray(scene.material_ids, dtype=np.uint32)
        self.tex_arrays = [np.ascontiguousarray(t, dtype=np.uint8) for t in scene.textures]
        self.textures = (RtTextureDesc * max(1, len(self.tex_arrays)))()
        for i, t in enumerate(self.tex_arrays):
            self.textures[i].width = t.shape[1]
            self.textures[i].height = t.shape[0]
            self.textures[i].rgba8 = u8ptr(t)
        self.materials = (RtMaterialDesc * max(1, len(scene.materials)))()
        for i, m in enumerate(scene.materials):
            d = self.materials[i]
            for k in range(4):
                d.color[k] = np.float32(m.color[k])
            e = m.emission_f32()
            for k in range(3):
                d.emission[k] = e[k]
            d.roughness = np.float32(m.roughness)
            d.metallic = np.float32(m.metallic)
            d.ior = np.float32(m.ior)
            d.color_tex = m.color_tex
            d.emissive_tex = m.emissive_tex
            d.metallic_roughness_tex = m.metallic_roughness_tex
            d.normal_tex = m.normal_tex
        d = RtSceneDesc()
        d.abi_version = RT_ABI_VERSION
        d.n_triangles = n
        d.positions = fptr(self.positions)
        d.normals = fptr(self.normals)
        d.texcoords = fptr(self.texcoords)
        d.tangents = fptr(self.tangents)
        d.material_ids = u32ptr(self.material_ids)
        d.n_materials = len(scene.materials)
        d.materials = self.materials
        d.n_textures = len(self.tex_arrays)
        d.textures = self.textures
        cam = scene.camera
        for k in range(3):
            d.camera.position[k] = np.float32(cam.position[k])
            d.camera.right[k] = np.float32(cam.right[k])
            d.camera.up[k] = np.float32(cam.up[k])
            d.camera.forward[k] = np.float32(cam.forward[k])
            d.bg_color[k] = np.float32(scene.bg_color[k])
        d.camera.fov_x = np.float32(cam.fov_x)
        d.ray_depth = scene.ray_depth
        prims = list(getattr(scene, "primitives", None) or [])
        self.primitives = (RtPrimitiveDesc * max(1, len(prims)))()
        for i, pr in enumerate(prims):  # (kind, material_id, param[3], position[3], rotation xyzw[4])
            q = self.primitives[i]
            q.kind, q.material_id = int(pr["kind"]), int(pr["material_id"])
            for k in range(3):
                q.param[k] = np.float32(pr["param"][k])
                q.position[k] = np.float32(pr["position"][k])
            for k in range(4):
                q.rotation[k] = np.float32(pr["rotation"][k])
        d.n_primitives = len(prims)
        d.primitives = self.primitives
        d.build_flags = int(getattr(scene, "build_flags", 0))
        for k, v in (getattr(scene, "build_options", None) or {}).items():  # rt_build_options fields by name
            setattr(d.build, k, v)
        d.bg_texture = int(getattr(scene, "bg_texture", -1))  # Scene::bg: index into textures, -1 = the white default
        self.desc = d


def desc_to_arrays(desc: RtSceneDesc) -> dict:
    """Copy an rt_scene_desc (e.g. from the C glTF loader) into numpy arrays."""
    n = desc.n_triangles

    def arr(ptr, count, dtype):
        if count == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)

    out = {
        "positions": arr(desc.positions, 9 * n, np.float32).reshape(n, 3, 3),
        "normals": arr(desc.normals, 9 * n, np.float32).reshape(n, 3, 3),
        "texcoords": arr(desc.texcoords, 6 * n, np.float32).reshape(n, 3, 2),
        "tangents": arr(desc.tangents, 9 * n, np.float32).reshape(n, 3, 3),
        "material_ids": arr(desc.material_ids, n, np.uint32),
        "materials": [],
        "textures": [],
        "camera": {
            "position": np.array(list(desc.camera.position), dtype=np.float32),
            "right": np.array(list(desc.camera.right), dtype=np.float32),
            "up": np.array(list(desc.camera.up), dtype=np.float32),
            "forward": np.array(list(desc.camera.forward), dtype=np.float32),
            "fov_x": np.float32(desc.camera.fov_x),
        },
        "bg_color": np.array(list(desc.bg_color), dtype=np.float32),
        "ray_depth": int(desc.ray_depth),
        "bg_texture": int(desc.bg_texture),
        "primitives": [
            {"kind": int(desc.primitives[i].kind), "material_id": int(desc.primitives[i].material_id), "param": np.array(list(desc.primitives[i].param), dtype=np.float32),
             "position": np.array(list(desc.primitives[i].position), dtype=np.float32), "rotation": np.array(list(desc.primitives[i].rotation), dtype=np.float32)}
            for i in range(desc.n_primitives)
        ],
    }
    for i in range(desc.n_materials):
        m = desc.materials[i]
        out["materials"].append(
            {
                "color": np.array(list(m.color), dtype=np.float32),
                "emission": np.array(list(m.emission), dtype=np.float32),
                "roughness": np.float32(m.roughness),
                "metallic": np.float32(m.metallic),
                "ior": np.float32(m.ior),
                "color_tex": m.color_tex,
                "emissive_tex": m.emissive_tex,
                "metallic_roughness_tex": m.metallic_roughness_tex,
                "normal_tex": m.normal_tex,
            }
        )
    for i in range(desc.n_textures):
        t = desc.textures[i]
        px = np.ctypeslib.as_array(t.rgba8, shape=(t.height, t.width, 4)).copy()
        out["textures"].append(px)
    return out

"""MI355X-native render loop for firelion9/raytracing-course-hw-public — Python host binding.

The product is the C-ABI shared library `csrc/librt_amd.so` (include/rt_abi.h + include/rt_host.h): hand-written
HIP kernels for gfx950 behind the seam the reference enters at `run_raytracer(scene, image)`
(src/raytracer.h:629). This module is a thin ctypes mirror of that ABI for tests and bench.py; names follow
the reference (`run_raytracer`, `parse_gltf_scene`, `Image.write`).

There is NO CPU fallback: if the library is missing, or no GPU is present when a device entry point is called,
the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

from . import scenegen  # noqa: F401
from ._ctypes_abi import (
    ABI_PROTOTYPES,
    ERROR_NAMES,
    HOST_PROTOTYPES,
    RT_FLAG_COUNTERS,
    RT_FLAG_DEVICE_FB,
    RT_BUILD_DEVICE_LBVH,
    RT_BUILD_WIDE,
    RT_BUILD_WIDE_HOST_COLLAPSE,
    RT_BUILD_LIGHTS_GLOBAL,
    RT_BUILD_GROUP_COPY,
    RT_BUILD_GROUP_SELF_EXCHANGE,
    RT_BUILDER_PLOC,
    RT_BUILDER_LBVH,
    RT_WIDE_ORDER_DEFAULT,
    RT_WIDE_ORDER_LEVEL,
    RT_WIDE_ORDER_DFS,
    RT_WIDE_ORDER_TREELET,
    RT_SORT_AUTO,
    RT_SORT_OFF,
    RT_SORT_CELL_OCTANT,
    RT_SORT_COARSE_CELL_DIR,
    RT_SORT_OCTANT_CELL,
    RT_SORT_CELL_OCTANT_CONE,
    RT_SORT_OCTANT_CELL_CONE,
    RT_SORT_OCTANT_FINE_CELL_CONE,
    RT_PACKET_AUTO,
    RT_PACKET_OFF,
    RT_PACKET_ON,
    RT_PROGRESS_FN,
    RT_FLAG_MEGAKERNEL,
    RT_FLAG_GLOBAL_BEST,
    RT_CAST_PROBE,
    RT_CAST_EXTEND,
    RT_CAST_EXTEND_GLOBAL,
    RT_CAST_PACKET,
    RT_CAST_PACKET_GLOBAL,
    RT_OK,
    RT_RNG_DEVICE,
    RT_RNG_REFERENCE,
    DescHolder,
    RtParams,
    RtSceneDesc,
    RtStats,
    bind,
    c_u8_p,
    desc_to_arrays,
    fptr,
    u8ptr,
    u32ptr,
)

RT_ALL_DEVICES = -1
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "csrc", "librt_amd.so")  # RT_AMD_LIB: tuning variants only
_lib: Optional[C.CDLL] = None


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {msg}")
        self.code = code


def lib() -> C.CDLL:
    """Load csrc/librt_amd.so. Raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: the HIP library is not built; there is no CPU fallback. Run __graft_entry__.build().")
        _lib = C.CDLL(LIB_PATH)
        bind(_lib, ABI_PROTOTYPES)
        bind(_lib, HOST_PROTOTYPES)
    return _lib


def _check(code: int) -> None:
    if code != RT_OK:
        raise RtError(code, lib().rt_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    return int(lib().rt_device_count())


class LoadedScene:
    """Result of parse_gltf_scene (scene.h:183) through the C++ host loader; owns the C-side arrays."""

    def __init__(self, handle: C.c_void_p):
        self._h = handle
        self.desc: RtSceneDesc = lib().rt_loaded_desc(handle).contents

    def arrays(self) -> dict:
        return desc_to_arrays(self.desc)

    def info(self) -> dict:
        """DIMENSIONS / SAMPLES of a scene-txt file (0 for glTF) and the number of ignored NEW_LIGHT blocks."""
        w, h, s, l = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().rt_loaded_info(self._h, C.byref(w), C.byref(h), C.byref(s), C.byref(l)))
        return {"width": w.value, "height": h.value, "samples": s.value, "ignored_lights": l.value}

    def set_env_map(self, image_path: str, intensity: float = 1.0) -> None:
        """main.cpp:28-31 under USE_ENV_MAP: scene.bg = Texture::load_img(image_path), bg_color = intensity (rt_loaded_set_env_map)."""
        _check(lib().rt_loaded_set_env_map(self._h, os.fsencode(image_path), C.c_float(intensity)))

    def disable_textures(self) -> None:
        """USE_TEXTURES = false (config.h:31-32): every texture lookup returns the texture's first texel (rt_loaded_disable_textures)."""
        _check(lib().rt_loaded_disable_textures(self._h))

    LIGHT_TRIANGLE_RELATIVE_POS = ((10.0, 0.0, -0.1), (0.0, 10.0, -0.1), (0.0, -10.0, -0.1))  # config.h:43-47

    def add_light_triangle(self, rel=LIGHT_TRIANGLE_RELATIVE_POS, intensity: float = 10.0) -> None:
        """scene.h:479-498 under ADD_LIGHT_TRIANGLE: an emissive triangle in the camera's frame (rt_loaded_add_light_triangle)."""
        r = np.ascontiguousarray(rel, dtype=np.float32).reshape(9)
        _check(lib().rt_loaded_add_light_triangle(self._h, fptr(r), C.c_float(intensity)))

    def close(self) -> None:
        if self._h:
            lib().rt_loaded_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse_gltf_scene(path: str, aspect: float) -> LoadedScene:
    h = C.c_void_p()
    _check(lib().rt_gltf_load(os.fsencode(path), C.c_float(aspect), C.byref(h)))
    return LoadedScene(h)


def parse_scene_txt(path: str) -> LoadedScene:
    """The scene-txt front end (csrc/host/txt_loader.cpp): sample_data-style scene files -> triangles + analytic primitives."""
    h = C.c_void_p()
    _check(lib().rt_txt_load(os.fsencode(path), C.byref(h)))
    return LoadedScene(h)


def load_scene(path: str, aspect: float) -> LoadedScene:
    """What the CLI does: scene-txt for *.txt, glTF otherwise (rt_scene_load)."""
    h = C.c_void_p()
    _check(lib().rt_scene_load(os.fsencode(path), C.c_float(aspect), C.byref(h)))
    return LoadedScene(h)


def _as_desc(scene) -> Tuple[RtSceneDesc, object]:
    if isinstance(scene, LoadedScene):
        return scene.desc, scene
    if isinstance(scene, RtSceneDesc):
        return scene, None
    holder = DescHolder(scene)
    return holder.desc, holder


def _apply_tuning(p: RtParams, tuning: dict):
    """rt_params' ABI-4 fields from keyword arguments; returns the ctypes callback object (if any), to be kept alive by the caller."""
    cb = None
    for k, v in tuning.items():
        if k == "progress":
            if v is not None:
                cb = RT_PROGRESS_FN(lambda done, total, user, f=v: f(done, total))
                p.progress = cb
        elif k in ("sort_mode", "packet_mode", "max_paths"):
            setattr(p, k, int(v))
        elif k == "packet_min_lanes":
            p.packet_min_lanes = float(v)
        else:
            raise TypeError(f"rt_params has no tuning field {k!r}")
    return cb


class DeviceScene:
    """Device-resident scene + both BVHs: the RaytracerStaticContext of raytracer.h:434-455, in HBM."""

    def __init__(self, scene, device=0, device_bvh: bool = False, wide: bool = False, build_flags: int = 0, **build_options):
        """`device_bvh`: build the scene BVH on the GPU (RT_BUILD_DEVICE_LBVH: production mode, different topology) instead
        of the reference-topology host build. `wide`: collapse that binary tree into the 8-wide quantised tree (RT_BUILD_WIDE:
        production traversal). `build_flags`: further RT_BUILD_* bits; `build_options`: rt_build_options fields by name (device_builder,
        ploc_radius, lbvh_leaf_tris, node_order, wide_cost_node, wide_cost_tri, wide_order). `device`: a HIP ordinal; RT_ALL_DEVICES (-1)
        for one replica per visible GPU + an RCCL communicator; or a list of ordinals (rt_create_on). Multi-GPU scenes shard every render
        over their GPUs and gather on the first one."""
        desc, keep = _as_desc(scene)
        self._keep = keep
        self._h = C.c_void_p()
        if device_bvh or wide or build_flags or build_options:  # a private copy of the descriptor with the build flags set
            d2 = RtSceneDesc()
            C.memmove(C.byref(d2), C.byref(desc), C.sizeof(RtSceneDesc))
            d2.build_flags = int(desc.build_flags) | (RT_BUILD_DEVICE_LBVH if device_bvh else 0) | (RT_BUILD_WIDE if wide else 0) | int(build_flags)
            for k, v in build_options.items():
                if not hasattr(d2.build, k):
                    raise TypeError(f"rt_build_options has no field {k!r}")
                setattr(d2.build, k, v)
            desc = d2
        if isinstance(device, (list, tuple)):
            devs = (C.c_int * len(device))(*[int(d) for d in device])
            _check(lib().rt_create_on(C.byref(desc), devs, len(device), C.byref(self._h)))
        else:
            _check(lib().rt_create(C.byref(desc), int(device), C.byref(self._h)))

    @property
    def n_devices(self) -> int:
        return int(lib().rt_scene_device_count(self._h))

    def close(self) -> None:
        if self._h:
            lib().rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_raytracer(
        self,
        width: int,
        height: int,
        samples: int,
        rng_mode: int = RT_RNG_DEVICE,
        seed: int = 0,
        shard_index: int = 0,
        shard_count: int = 1,
        shard_block: int = 0,
        out: Optional[np.ndarray] = None,
        device_fb: int = 0,
        counters: bool = False,
        megakernel: bool = False,
        global_best: bool = False,
        **tuning,
    ):
        """run_raytracer(scene, image) (raytracer.h:629): returns (linear float framebuffer (H,W,3), stats dict).
        With `device_fb` (a device pointer) the framebuffer stays in HBM and None is returned for it.
        `counters=True` runs the instrumented kernel variant and fills the event counters of the stats.
        `tuning`: rt_params fields by name — sort_mode (RT_SORT_*), packet_mode (RT_PACKET_*), packet_min_lanes, max_paths,
        progress (a callable (done, total))."""
        p = RtParams(width, height, samples, rng_mode, seed, shard_index, shard_count, shard_block,
                     (RT_FLAG_COUNTERS if counters else 0) | (RT_FLAG_MEGAKERNEL if megakernel else 0) | (RT_FLAG_GLOBAL_BEST if global_best else 0))
        keep_cb = _apply_tuning(p, tuning)  # noqa: F841 (keeps the ctypes callback alive for the call)
        st = RtStats()
        if device_fb:
            p.flags |= RT_FLAG_DEVICE_FB
            _check(lib().rt_render(self._h, C.byref(p), C.c_void_p(device_fb), C.byref(st)))
            return None, st.as_dict()
        fb = out if out is not None else np.zeros((height, width, 3), dtype=np.float32)
        assert fb.dtype == np.float32 and fb.flags["C_CONTIGUOUS"] and fb.size == width * height * 3
        _check(lib().rt_render(self._h, C.byref(p), fb.ctypes.data_as(C.c_void_p), C.byref(st)))
        return fb, st.as_dict()

    def run_raytracer_rgb8(
        self,
        width: int,
        height: int,
        samples: int,
        seed: int = 0,
        shard_index: int = 0,
        shard_count: int = 1,
        shard_block: int = 0,
        out: Optional[np.ndarray] = None,
        device_rgb8: int = 0,
        rng_mode: int = RT_RNG_DEVICE,
        global_best: bool = False,
        **tuning,
    ):
        """run_raytracer(scene, image) with the reference's own output type (image.h:40-42): the tone-mapped rgb8 image,
        film applied on the device. Returns ((H,W,3) uint8 array or None with `device_rgb8`, stats dict). `tuning`: as run_raytracer."""
        p = RtParams(width, height, samples, rng_mode, seed, shard_index, shard_count, shard_block, RT_FLAG_GLOBAL_BEST if global_best else 0)
        keep_cb = _apply_tuning(p, tuning)  # noqa: F841
        st = RtStats()
        if device_rgb8:
            p.flags |= RT_FLAG_DEVICE_FB
            _check(lib().rt_render_rgb8(self._h, C.byref(p), C.c_void_p(device_rgb8), C.byref(st)))
            return None, st.as_dict()
        img = out if out is not None else np.zeros((height, width, 3), dtype=np.uint8)
        assert img.dtype == np.uint8 and img.flags["C_CONTIGUOUS"] and img.size == width * height * 3
        _check(lib().rt_render_rgb8(self._h, C.byref(p), img.ctypes.data_as(C.c_void_p), C.byref(st)))
        return img, st.as_dict()

    def film_rgb8(self, fb: np.ndarray) -> np.ndarray:
        """The device film (image.h:49-82 on the GPU) applied to a host float array of shape (..., 3)."""
        fb = np.ascontiguousarray(fb, dtype=np.float32)
        out = np.zeros(fb.shape, dtype=np.uint8)
        _check(lib().rt_film_rgb8(self._h, fptr(fb), fb.size // 3, u8ptr(out)))
        return out

    def cast_rays(self, rays: np.ndarray):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        prim = np.zeros(n, dtype=np.uint32)
        bct = np.zeros((n, 3), dtype=np.float32)
        _check(lib().rt_cast_rays(self._h, fptr(rays), n, u32ptr(prim), fptr(bct)))
        return prim, bct

    def cast_rays_ex(self, rays: np.ndarray, mode: int):
        """rt_cast_rays_ex: the closest-hit probe through the renderer's own kernels (mode = RT_CAST_*).
        Returns (prim, bct, stats dict with casts / nodes_visited / box_tests / tri_tests / kernel_ms)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        prim = np.zeros(n, dtype=np.uint32)
        bct = np.zeros((n, 3), dtype=np.float32)
        st = RtStats()
        _check(lib().rt_cast_rays_ex(self._h, fptr(rays), n, int(mode), u32ptr(prim), fptr(bct), C.byref(st)))
        return prim, bct, st.as_dict()

    def surface_normals(self, rays: np.ndarray):
        """rt_surface_normals: closest hit + the normals to_intersection_info (bvh.h:80-121) hands to shade().
        Returns (prim, t, normal (n,3), shading_normal (n,3))."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        prim = np.zeros(n, dtype=np.uint32)
        t = np.zeros(n, dtype=np.float32)
        nn = np.zeros((n, 3), dtype=np.float32)
        sn = np.zeros((n, 3), dtype=np.float32)
        _check(lib().rt_surface_normals(self._h, fptr(rays), n, u32ptr(prim), fptr(t), fptr(nn), fptr(sn)))
        return prim, t, nn, sn

    def light_pdf(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        out = np.zeros(n, dtype=np.float32)
        _check(lib().rt_light_pdf(self._h, fptr(rays), n, fptr(out)))
        return out

    def bg_at(self, dirs: np.ndarray) -> np.ndarray:
        """Scene::bg_at (scene.h:83-89) for explicit directions -> (n, 3) rgb (rt_bg_at)."""
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = dirs.shape[0]
        out = np.zeros((n, 3), dtype=np.float32)
        _check(lib().rt_bg_at(self._h, fptr(dirs), n, fptr(out)))
        return out

    def bvh_device_dump(self, which: int = 0):
        """The BVH as the kernels see it, read back from HBM: {root, nodes (n_inner,16) u32, tris (n_tris,12) u32}."""
        ni, nt, root = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().rt_bvh_device_dump(self._h, which, C.byref(ni), C.byref(nt), C.byref(root), None, None))
        nodes = np.zeros((ni.value, 16), dtype=np.uint32)
        tris = np.zeros((nt.value, 12), dtype=np.uint32)
        _check(lib().rt_bvh_device_dump(self._h, which, C.byref(ni), C.byref(nt), C.byref(root), u32ptr(nodes), u32ptr(tris)))
        return {"root": root.value, "nodes": nodes, "tris": tris}

    def bvh_wide_dump(self):
        """The 8-wide scene BVH read back from HBM: {depth, nodes (n,20) u32 (WideNode records), tris (n_tris,12) u32}."""
        nn, nt, dp = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().rt_bvh_wide_dump(self._h, C.byref(nn), C.byref(nt), C.byref(dp), None, None))
        nodes = np.zeros((nn.value, 20), dtype=np.uint32)
        tris = np.zeros((nt.value, 12), dtype=np.uint32)
        _check(lib().rt_bvh_wide_dump(self._h, C.byref(nn), C.byref(nt), C.byref(dp), u32ptr(nodes), u32ptr(tris)))
        return {"depth": dp.value, "nodes": nodes, "tris": tris}

    def build_times(self):
        b, u, w = C.c_double(), C.c_double(), C.c_double()
        _check(lib().rt_build_times_ex(self._h, C.byref(b), C.byref(u), C.byref(w)))
        return {"build_ms": b.value, "upload_ms": u.value, "wide_ms": w.value}

    def bvh_info(self, which: int):
        nn, no, root = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().rt_bvh_info(self._h, which, C.byref(nn), C.byref(no), C.byref(root), None, None))
        nodes = np.zeros((nn.value, 10), dtype=np.uint32)
        order = np.zeros(no.value, dtype=np.uint32)
        _check(lib().rt_bvh_info(self._h, which, C.byref(nn), C.byref(no), C.byref(root), u32ptr(nodes), u32ptr(order)))
        return {"root": root.value, "nodes": nodes, "order": order}


def bvh_build_host(positions: np.ndarray, subset: Optional[np.ndarray] = None) -> dict:
    """BVH::build (bvh.h:368-393) with the library's host builder, no GPU needed: {root, nodes (n,10) u32, order}."""
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 9)
    n = pos.shape[0]
    sub = np.arange(n, dtype=np.uint32) if subset is None else np.ascontiguousarray(subset, dtype=np.uint32)
    nodes = np.zeros((2 * len(sub) + 1, 10), dtype=np.uint32)
    order = np.zeros(len(sub), dtype=np.uint32)
    nn, root = C.c_uint32(), C.c_uint32()
    rc = lib().rt_bvh_build_host(fptr(pos), n, u32ptr(sub), len(sub), C.byref(nn), C.byref(root), u32ptr(nodes), u32ptr(order))
    if rc != RT_OK:
        raise RtError(rc, "rt_bvh_build_host")
    return {"root": root.value, "nodes": nodes[: nn.value].copy(), "order": order}


def bvh_wide_build_host(positions: np.ndarray, cost_node: float = 1.0, cost_tri: float = 0.3) -> dict:
    """The production build (RT_BUILD_WIDE) on the host, no GPU needed: {nodes (n,20) u32 WideNode records, order, depth, sah_cost}."""
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 9)
    n = pos.shape[0]
    nn, depth, cost = C.c_uint32(), C.c_uint32(), C.c_double()
    rc = lib().rt_bvh_wide_build_host(fptr(pos), n, cost_node, cost_tri, C.byref(nn), C.byref(depth), C.byref(cost), None, 0, None)
    if rc != RT_OK:
        raise RtError(rc, "rt_bvh_wide_build_host")
    nodes = np.zeros((nn.value, 20), dtype=np.uint32)
    order = np.zeros(n, dtype=np.uint32)
    rc = lib().rt_bvh_wide_build_host(fptr(pos), n, cost_node, cost_tri, C.byref(nn), C.byref(depth), C.byref(cost), u32ptr(nodes), nn.value, u32ptr(order))
    if rc != RT_OK:
        raise RtError(rc, "rt_bvh_wide_build_host")
    return {"nodes": nodes, "order": order, "depth": depth.value, "sah_cost": cost.value}


def tonemap(fb: np.ndarray) -> np.ndarray:
    """Image::set_pixel's convert_color (image.h:40-42, 79-82) over a whole linear framebuffer -> (H,W,3) u8."""
    fb = np.ascontiguousarray(fb, dtype=np.float32)
    out = np.zeros(fb.shape, dtype=np.uint8)
    lib().rt_tonemap_rgb8(fptr(fb), fb.size // 3, u8ptr(out))
    return out


def film_table():
    """(thr[256] float32, special[3] uint32): the verified gamma + quantise thresholds the device film searches."""
    thr = np.zeros(256, dtype=np.float32)
    special = np.zeros(3, dtype=np.uint32)
    _check(lib().rt_film_table(fptr(thr), u32ptr(special)))
    return thr, special


def write_ppm(path: str, rgb8: np.ndarray) -> None:
    """Image::write (image.h:34-38)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    _check(lib().rt_write_ppm(os.fsencode(path), w, h, u8ptr(rgb8)))


def image_decode(path: str) -> np.ndarray:
    """Texture::load_img (geometry.h:584-598) for PNG and JPEG files: (H, W, 4) uint8, the bytes stb_image returns."""
    w, h = C.c_uint32(), C.c_uint32()
    p = c_u8_p()
    _check(lib().rt_image_decode_file(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p)))
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        lib().rt_free(p)


def png_decode(path: str) -> np.ndarray:
    w, h = C.c_uint32(), C.c_uint32()
    p = c_u8_p()
    _check(lib().rt_png_decode_file(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p)))
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        lib().rt_free(p)


__all__ = [
    "DeviceScene",
    "LoadedScene",
    "RT_ALL_DEVICES",
    "RT_RNG_DEVICE",
    "RT_RNG_REFERENCE",
    "RtError",
    "device_count",
    "lib",
    "parse_gltf_scene",
    "parse_scene_txt",
    "load_scene",
    "png_decode",
    "scenegen",
    "tonemap",
    "write_ppm",
]

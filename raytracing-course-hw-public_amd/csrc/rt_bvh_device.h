// rt_bvh_device.h — device-side BVH construction (rt_bvh_device.hip); internal.
#pragma once
#include <hip/hip_runtime_api.h>

#include "../../include/rt_abi.h"
#include "rt_device_types.h"

namespace rt {

struct DeviceBvh {
    DevNode *nodes = nullptr; // device memory, owned by the caller after a successful build (null when `wide` was built)
    WideNode *wide = nullptr; // the 8-wide tree, collapsed on the device (tris / attrs are then in ITS order)
    uint32_t n_wide = 0, wide_depth = 0;
    WideGrid wide_grid{};     // the grids its nodes were snapped to (wide_grid.h)
    double wide_ms = 0;
    bool ploc = false;        // the binary tree came from PLOC (else: Karras radix tree)
    uint32_t rounds = 0;      // PLOC rounds
    DevTri *tris = nullptr;
    DevAttr *attrs = nullptr;
    uint32_t n_inner = 0, n_tris = 0, root = RT_NONE;
    bool fast_ok = true;      // every vertex coordinate is 0 or within [2^-37, 2^40] in magnitude (div_exact_fast)
    float lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
    double upload_ms = 0, build_ms = 0;
};

// Linear BVH over all triangles of `d`, built on the current device on `stream` (blocking). On failure *err names the call.
// `wide`: also collapse it into the 8-wide quantised tree ON THE DEVICE (the dynamic program of wide_build.cpp inside the refit,
// then a top-down emission, level by level) and keep only that; ignored for scenes of <= 8 triangles (out->wide stays null).
hipError_t build_bvh_device(const rt_scene_desc *d, hipStream_t stream, DeviceBvh *out, const char **err, bool wide = false, float cost_node = 1.0f,
                            float cost_tri = 0.3f);

} // namespace rt

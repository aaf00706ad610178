// rt_wide_pack.h — WideNode[] + DevTri[] (device memory, either builder's output) -> the packed blob the wide kernels read (rt_wide_pack.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include "rt_device_types.h"

namespace rt {
// Blocking (synchronises `stream`). On success *blob_out is a device allocation of *n_units_out 16-byte units owned by the caller (null for an
// empty tree); on failure *err names the call. `grid` must be the grid the builder snapped its nodes to (WideBvh::grid / DeviceBvh::wide_grid).
hipError_t pack_wide_device(const WideNode *d_nodes, uint32_t n_wide, const DevTri *d_tris, const WideGrid &grid, hipStream_t stream, uint4_pod **blob_out,
                            uint32_t *n_units_out, const char **err);
} // namespace rt

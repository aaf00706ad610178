// rt_kernels.h — launchers of the HIP kernels in rt_kernels.hip (internal; the public boundary is include/rt_abi.h).
#pragma once
#include <hip/hip_runtime_api.h>

#include "rt_device_types.h"

namespace rt {
hipError_t launch_render(const DevScene &S, const RenderLaunch &L, bool stats, int blocks, hipStream_t stream);
hipError_t launch_cast(const DevScene &S, const float *rays, uint32_t n, uint32_t *prim, float *bct, hipStream_t stream);
hipError_t launch_light_pdf(const DevScene &S, const float *rays, uint32_t n, float *pdf, hipStream_t stream);
} // namespace rt

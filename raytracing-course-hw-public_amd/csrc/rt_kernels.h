// rt_kernels.h — launchers of the HIP kernels (internal; the public boundary is include/rt_abi.h).
#pragma once
#include <hip/hip_runtime_api.h>

#include <vector>

#include "rt_device_types.h"

namespace rt {
// rt_kernels.hip: persistent megakernel (reference-RNG parity mode; cross-check of the wavefront path) + probes
hipError_t launch_render(const DevScene &S, const RenderLaunch &L, bool stats, int blocks, hipStream_t stream);
hipError_t launch_cast(const DevScene &S, const float *rays, uint32_t n, uint32_t *prim, float *bct, hipStream_t stream);
hipError_t launch_light_pdf(const DevScene &S, const float *rays, uint32_t n, float *pdf, hipStream_t stream);
// rt_wavefront.hip: one pass (pixel tile x sample range) of the wavefront pipeline, stream-ordered
// `extend_events` (optional): receives one (start, stop) event pair per wf_extend launch, recorded on `stream`
hipError_t launch_wavefront_pass(const DevScene &S, WfLaunch L, bool stats, int num_cus, bool first_pass, bool last_pass, hipStream_t stream,
                                 std::vector<hipEvent_t> *extend_events);
// bytes of temporary storage rocPRIM's radix sort needs for `n` (key, slot) pairs
size_t wavefront_sort_temp_bytes(size_t n);
} // namespace rt

// rt_kernels.h — launchers of the HIP kernels (internal; the public boundary is include/rt_abi.h).
#pragma once
#include <hip/hip_runtime_api.h>

#include <vector>

#include "rt_device_types.h"

// Launch + check. hipGetLastError() reports the last error of ANY earlier HIP call of this thread (e.g. a refused
// hipSetDevice in another scene's rt_create), so the sticky state is cleared first: what comes back belongs to this launch.
#define RT_LAUNCH_CHECKED(...)               \
    ({                                       \
        (void)hipGetLastError();             \
        hipLaunchKernelGGL(__VA_ARGS__);     \
        hipGetLastError();                   \
    })

namespace rt {
// HIP events for per-launch timing, created once per scene and reused by every render (no create/destroy inside the
// timed region). next() returns nullptr when an event cannot be created; pairs are taken in (start, stop) order.
struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    hipEvent_t next() {
        if (used == ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess)
                return nullptr;
            ev.push_back(e);
        }
        return ev[used++];
    }
    void reset() { used = 0; }
    void destroy() {
        for (hipEvent_t e : ev)
            (void)hipEventDestroy(e);
        ev.clear();
        used = 0;
    }
};
// rt_kernels.hip: persistent megakernel (reference-RNG parity mode; cross-check of the wavefront path) + probes
hipError_t launch_render(const DevScene &S, const RenderLaunch &L, bool stats, int blocks, hipStream_t stream);
hipError_t launch_cast(const DevScene &S, const float *rays, uint32_t n, uint32_t *prim, float *bct, hipStream_t stream);
hipError_t launch_surface_normals(const DevScene &S, const float *rays, uint32_t n, uint32_t *prim, float *t, float *normal, float *shading, hipStream_t stream);
hipError_t launch_light_pdf(const DevScene &S, const float *rays, uint32_t n, float *pdf, hipStream_t stream);
hipError_t launch_bg_at(const DevScene &S, const float *dirs, uint32_t n, float *rgb, hipStream_t stream);
// rt_wavefront.hip: one pass (pixel tile x sample range) of the wavefront pipeline, stream-ordered
// `extend_events` (optional): one (start, stop) event pair per wf_extend launch is taken from the pool and recorded on `stream`
// `packet_census_out` (optional, host, 2 words): trips and lanes served of this pass's wf_extend_packet launch (0, 0 if it did not run)
// `host_sync` (optional): pinned words + events for the one-bounce-late read-back of the queue sizes (coherence sort and the
// early stop need an upper bound of the queue on the host); without it bounces >= 1 run unsorted over the full pass
struct WfHostSync {
    uint32_t *counts;    // pinned host memory: word b = size of the queue entering bounce b; words WF_HOST_CENSUS_WORD.. = packet census
    hipEvent_t *events;  // events[b] is recorded once counts[b] has been written
    int n_events;
};
#define WF_HOST_CENSUS_WORD 40 /* 8-byte aligned, behind RT_MAX_RAY_DEPTH + 1 size words */
hipError_t launch_wavefront_pass(const DevScene &S, WfLaunch L, bool stats, int num_cus, bool first_pass, bool last_pass, hipStream_t stream,
                                 EventPool *extend_events, unsigned long long *packet_census_out, const WfHostSync *host_sync);
// closest-hit probe through the renderer's own kernels: `rays` (6 floats each, device) -> queue -> wf_extend (or wf_extend_packet)
// -> prim / bct (device). `L` carries the workspace (paths_in, hits, counters, stack_overflow, stats) and the traversal mode.
hipError_t launch_wavefront_cast(const DevScene &S, WfLaunch L, const float *rays, uint32_t n, bool packet, bool stats, uint32_t *prim, float *bct,
                                 hipStream_t stream);
// rt_wide.hip: the closest-hit kernel of scenes built with RT_BUILD_WIDE (same queue protocol as wf_extend)
// `packet`: the wave walks the tree once for its 64 consecutive rays (coherent primary rays), records through the scalar cache
hipError_t launch_extend_wide(const DevScene &S, const WfLaunch &L, bool packet, bool stats, int blocks, hipStream_t stream);
// bytes of temporary storage rocPRIM's radix sort needs for `n` (key, slot) pairs
size_t wavefront_sort_temp_bytes(size_t n);
} // namespace rt

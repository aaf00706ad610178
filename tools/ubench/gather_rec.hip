// gather_rec.hip — what a node record's SIZE and ALIGNMENT cost a per-lane gather (round 4: the wide traversal's L1-miss stream sits at ~97 G
// requests/s on both bench scenes, 0.85 of what gather64.hip measures for an L2-resident table: that request rate is its roof, so the question is
// how many requests one node visit makes). Every lane reads LOADS 16-byte pieces of one pseudo-random record per step:
//   stride 64, 4 loads   a 64-byte node, 64-byte aligned: never crosses a 128-byte line, two nodes per line
//   stride 80, 5 loads   today's WideNode: half of all records cross a line
//   stride 128, 5 loads  the same 80 bytes padded to a line of their own
//   stride 48, 3 loads   a DevTri record
// over tables that fit the L2s (16 MiB: 2 MiB per XCD), the Infinity Cache (128 MiB) and neither (1 GiB). Dependent chain (a BVH descent).
//   hipcc --offload-arch=gfx950 -O3 gather_rec.hip -o gather_rec && ./gather_rec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}
template <int STRIDE, int LOADS> __global__ __launch_bounds__(256) void k_gather(const char *__restrict__ tab, uint32_t n_rec, uint32_t iters, uint32_t *out) {
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x + 0x9E3779B9u);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t r = (uint32_t)(((unsigned long long)idx * n_rec) >> 32);
        const uint4 *p = reinterpret_cast<const uint4 *>(tab + (size_t)r * STRIDE);
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const uint4 a = p[k];
            v ^= a.x ^ a.w;
        }
        acc += v;
        idx = mix(idx + it + v); // zero-filled table: v == 0, but the compiler cannot know
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}
template <int STRIDE, int LOADS> static void run(const char *tab, size_t bytes, uint32_t *o, hipEvent_t e0, hipEvent_t e1) {
    const int blocks = 256 * 8;
    const uint32_t iters = 256;
    const uint32_t n_rec = (uint32_t)(bytes / STRIDE);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_gather<STRIDE, LOADS>), dim3(blocks), dim3(256), 0, 0, tab, n_rec, iters, o);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best)
            best = ms;
    }
    const double recs = (double)blocks * 256 * iters;
    printf("table %5zu MiB, stride %3d B, %d x 16-B loads: %.3f ms, %6.2f Grec/s\n", bytes >> 20, STRIDE, LOADS, best, recs / best / 1e6);
}
int main() {
    uint32_t *o;
    if (hipMalloc(&o, 4) != hipSuccess)
        return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (size_t mib : {16, 128, 1024}) {
        const size_t bytes = mib << 20;
        char *tab;
        if (hipMalloc(&tab, bytes + 256) != hipSuccess)
            return 1;
        hipMemset(tab, 0, bytes + 256);
        hipDeviceSynchronize();
        run<64, 4>(tab, bytes, o, e0, e1);
        run<80, 5>(tab, bytes, o, e0, e1);
        run<128, 5>(tab, bytes, o, e0, e1);
        run<48, 3>(tab, bytes, o, e0, e1);
        run<64, 3>(tab, bytes, o, e0, e1);
        hipFree(tab);
    }
    return 0;
}

// leaf_gather.hip — does the vector L1 charge one tag access per LANE or per distinct line when adjacent lanes read neighbouring records? The parity
// kernel's leaf batches put the triangles of one leaf on adjacent lanes (DevTri: 48-byte records, three 16-byte loads per lane). Two layouts of the
// same bytes, groups of 4 records at pseudo-random group positions, every lane of a quad reads "its" record of the quad's group:
//   AoS  record t of the group at  group * 192 + t * 48,            piece p at + 16 p     (today's DevTri[])
//   SoA  piece p of record t at    group * 192 + p * 64 + t * 16                          (each load instruction of a quad covers 64 contiguous bytes)
//   hipcc --offload-arch=gfx950 -O3 leaf_gather.hip -o leaf_gather && ./leaf_gather
//   rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d out -- ./leaf_gather
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}
template <bool SOA> __global__ __launch_bounds__(256) void k_leaf(const char *__restrict__ tab, uint32_t n_groups, uint32_t iters, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u, t = lane & 3u;
    uint32_t idx = mix((blockIdx.x * 256u + threadIdx.x) / 4u + 0x9E3779B9u); // one group per quad
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t g = (uint32_t)(((unsigned long long)idx * n_groups) >> 32);
        const char *base = tab + (size_t)g * 192u;
        uint4 a, b, c;
        if (SOA) {
            a = *reinterpret_cast<const uint4 *>(base + t * 16u);
            b = *reinterpret_cast<const uint4 *>(base + 64u + t * 16u);
            c = *reinterpret_cast<const uint4 *>(base + 128u + t * 16u);
        } else {
            a = *reinterpret_cast<const uint4 *>(base + t * 48u);
            b = *reinterpret_cast<const uint4 *>(base + t * 48u + 16u);
            c = *reinterpret_cast<const uint4 *>(base + t * 48u + 32u);
        }
        const uint32_t v = a.x ^ b.y ^ c.z;
        acc += v;
        idx = mix(idx + it + __shfl((int)v, (int)(lane & ~3u))); // the quad moves on together
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}
int main() {
    uint32_t *o;
    if (hipMalloc(&o, 4) != hipSuccess)
        return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (size_t mib : {12, 96, 768}) {
        const size_t bytes = mib << 20;
        char *tab;
        if (hipMalloc(&tab, bytes + 256) != hipSuccess)
            return 1;
        hipMemset(tab, 0, bytes + 256);
        hipDeviceSynchronize();
        for (int soa = 0; soa < 2; ++soa) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (soa)
                    hipLaunchKernelGGL((k_leaf<true>), dim3(2048), dim3(256), 0, 0, tab, (uint32_t)(bytes / 192), 256u, o);
                else
                    hipLaunchKernelGGL((k_leaf<false>), dim3(2048), dim3(256), 0, 0, tab, (uint32_t)(bytes / 192), 256u, o);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best)
                    best = ms;
            }
            const double recs = 2048.0 * 256 * 256;
            printf("table %4zu MiB, %s: %.3f ms, %.2f G records/s\n", mib, soa ? "SoA (64 contiguous bytes per quad and load)" : "AoS (48-byte records)                     ", best, recs / best / 1e6);
        }
        hipFree(tab);
    }
    return 0;
}

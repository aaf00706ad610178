// gather64.hip — calibrates the two numbers the S-10M roofline needs (MI355X_MICROARCH.md: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern"):
//   (1) how many bytes FETCH_SIZE reports per RANDOM 64-byte record read from a table far larger than the caches (the access
//       pattern of wf_extend on a DevNode[] of 0.64 GB: one 64-B record per lane per step, no reuse inside a wave);
//   (2) the rate at which the chip sustains such gathers (records/s and useful GB/s), i.e. the practical memory roof for it.
// Every lane reads REC bytes (64 or 128, as 16-B loads) at a pseudo-random record index; `loads in flight per lane` = 1
// (dependent chain like a BVH descent: the next index depends on the loaded data) or 4 (independent).
//   hipcc --offload-arch=gfx950 -O3 gather64.hip -o gather64 && ./gather64 [table MiB ...]
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- ./gather64     (counter per kernel launch; records per launch printed)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}
template <int REC, bool DEPENDENT> __global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ tab, uint32_t n_rec_mask, uint32_t iters, uint32_t *out) {
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x + 0x9E3779B9u);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint4 *p = tab + (size_t)(idx & n_rec_mask) * (REC / 16);
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        uint32_t v = a.x ^ b.y ^ c.z ^ d.w;
        if (REC == 128) {
            uint4 e = p[4], f = p[5], g = p[6], h = p[7];
            v ^= e.x ^ f.y ^ g.z ^ h.w;
        }
        acc += v;
        idx = mix(idx + it + (DEPENDENT ? v : 0u)); // table is zero-filled: v == 0, but the compiler cannot know
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}
int main(int argc, char **argv) {
    // table sizes in MiB (powers of two); default: the footprints that matter for wf_extend (S-10M DevNode[] = 640 MB) up to
    // far beyond every cache and TLB reach
    std::vector<size_t> sizes;
    for (int i = 1; i < argc; ++i)
        sizes.push_back((size_t)atoll(argv[i]));
    if (sizes.empty())
        sizes = {512, 1024, 2048, 8192};
    uint32_t *o;
    if (hipMalloc(&o, 4) != hipSuccess)
        return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int blocks = 256 * 8; // 8 blocks of 256 threads per CU: 32 waves per CU
    const uint32_t iters = 256;
    for (size_t mib : sizes) {
        const size_t bytes = mib << 20;
        uint4 *tab;
        if (hipMalloc(&tab, bytes) != hipSuccess)
            return 1;
        hipMemset(tab, 0, bytes);
        hipDeviceSynchronize();
        for (int variant = 0; variant < 4; ++variant) {
            const int rec = (variant & 1) ? 128 : 64;
            const bool dep = variant < 2;
            const uint32_t mask = (uint32_t)(bytes / rec - 1);
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (rec == 64 && dep)
                    hipLaunchKernelGGL((k_gather<64, true>), dim3(blocks), dim3(256), 0, 0, tab, mask, iters, o);
                else if (rec == 128 && dep)
                    hipLaunchKernelGGL((k_gather<128, true>), dim3(blocks), dim3(256), 0, 0, tab, mask, iters, o);
                else if (rec == 64)
                    hipLaunchKernelGGL((k_gather<64, false>), dim3(blocks), dim3(256), 0, 0, tab, mask, iters, o);
                else
                    hipLaunchKernelGGL((k_gather<128, false>), dim3(blocks), dim3(256), 0, 0, tab, mask, iters, o);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best)
                    best = ms;
            }
            const double recs = (double)blocks * 256 * iters;
            printf("table %5zu MiB, record %3d B, %s chain: %.0f records per launch, %.3f ms, %.2f Grec/s, %.0f GB/s useful\n", mib, rec, dep ? "dependent  " : "independent", recs, best,
                   recs / best / 1e6, recs * rec / best / 1e6);
        }
        hipFree(tab);
    }
    return 0;
}

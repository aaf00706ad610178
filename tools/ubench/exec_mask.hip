// exec_mask.hip — does a VALU instruction cost less when only some 16-lane groups of the wave are active? (gfx950)
// hipcc --offload-arch=gfx950 -O3 exec_mask.hip -o exec_mask && ./exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 65536
__global__ void k(float *out, unsigned long long mask, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const bool on = (mask >> (threadIdx.x & 63u)) & 1ull;
    if (on) {
        for (int i = 0; i < N_IT; ++i) {
            asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %4\n"
                         "v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    float *out;
    hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const unsigned long long masks[] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x1ull, 0x0001000100010001ull, 0x00FF00FF00FF00FFull, 0xFFFF0000FFFF0000ull};
    const char *names[] = {"all 64", "low 32", "low 16", "lane 0", "1 lane in each 16", "8 lanes in each 16", "groups 1 and 3"};
    for (int w = 4; w <= 8; w += 4)
        for (int m = 0; m < 7; ++m) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(64 * w), 0, 0, out, masks[m], 1.0f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("waves/CU %d  %-20s %.3f ms\n", w, names[m], best);
        }
    return 0;
}

// hbm_stream.hip — measured HBM bandwidth of the box (report nominal AND measured peak next to the roofline).
// copy (1 read + 1 write) and triad (2 reads + 1 write) over 2 GiB arrays, float4 per lane, grid-stride.
// hipcc --offload-arch=gfx950 -O3 hbm_stream.hip -o hbm_stream && ./hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ a, float4 *__restrict__ c, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        c[i] = a[i];
}
__global__ __launch_bounds__(256) void k_triad(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ c, size_t n, float s) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = a[i], y = b[i];
        c[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ a, float *out, size_t n) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = a[i];
        acc += x.x + x.y + x.z + x.w;
    }
    if (acc == 12345.678f)
        out[0] = acc;
}
int main() {
    const size_t bytes = 2ull << 30, n = bytes / sizeof(float4);
    float4 *a, *b, *c;
    float *o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&c, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess)
        return 1;
    hipMemset(a, 0, bytes), hipMemset(b, 0, bytes), hipMemset(c, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int grids[] = {256 * 8, 256 * 16, 256 * 32};
    for (int g : grids) {
        float best[3] = {1e9f, 1e9f, 1e9f};
        for (int rep = 0; rep < 6; ++rep)
            for (int k = 0; k < 3; ++k) {
                hipEventRecord(e0);
                if (k == 0)
                    hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, c, n);
                else if (k == 1)
                    hipLaunchKernelGGL(k_triad, dim3(g), dim3(256), 0, 0, a, b, c, n, 1.5f);
                else
                    hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, o, n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best[k])
                    best[k] = ms;
            }
        printf("grid %5d blocks: copy %.0f GB/s  triad %.0f GB/s  read %.0f GB/s\n", g, 2.0 * bytes / best[0] / 1e6, 3.0 * bytes / best[1] / 1e6, 1.0 * bytes / best[2] / 1e6);
    }
    return 0;
}

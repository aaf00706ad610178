// rt_film.hip — the film on the device: Image::set_pixel's tone mapping (reference src/image.h:49-82) for the pixels
// a shard rendered, so that the render call hands back (and a multi-GPU job gathers) rgb8 instead of float3.
//   aces(x) = (x*(a*x+b)) / (x*(c*x+d)+e)     image.h:51-59   five IEEE binary32 operations, no contraction
//   level   = round(clamp(powf(aces, 1/2.2f) * 255, 0, 255))   image.h:61-82
// powf is not re-implemented: the host builds, from its own libm, the 255 thresholds of the (monotone) map
// aces -> level and verifies them (host/film.cpp film_table); the kernel binary-searches that table. HBM-bound
// streaming kernel: 12 B read + 3 B written per pixel, one thread per pixel.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "rt_film.h"

namespace {

__device__ __forceinline__ uint32_t film_level(float x, const float *thr, uint32_t lv_nan, uint32_t lv_neg, uint32_t lv_ninf) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    const float y = (x * (a * x + b)) / (x * (c * x + d) + e);
    if (!(y >= 0.0f)) { // NaN or negative (-0 compares equal to 0 and is looked up: level 0)
        if (y != y)
            return lv_nan;
        return y == -__builtin_inff() ? lv_ninf : lv_neg;
    }
    uint32_t lo = 0, hi = 256; // largest k with thr[k] <= y
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t mid = (lo + hi) >> 1;
        if (y >= thr[mid])
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void film_kernel(const float *__restrict__ fb, uint8_t *__restrict__ out, uint32_t n_pixels, uint32_t shard_index,
                                                   uint32_t shard_count, uint32_t shard_block, const rt::FilmTable *__restrict__ table) {
    __shared__ float s_thr[256];
    s_thr[threadIdx.x] = table->thr[threadIdx.x];
    const uint32_t lv_nan = table->special[0], lv_neg = table->special[1], lv_ninf = table->special[2];
    __syncthreads();
    for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n_pixels; p += gridDim.x * 256u) {
        if (shard_count > 1 && (p / shard_block) % shard_count != shard_index)
            continue;
        const float r = fb[3 * (size_t)p], g = fb[3 * (size_t)p + 1], b = fb[3 * (size_t)p + 2];
        out[3 * (size_t)p] = (uint8_t)film_level(r, s_thr, lv_nan, lv_neg, lv_ninf);
        out[3 * (size_t)p + 1] = (uint8_t)film_level(g, s_thr, lv_nan, lv_neg, lv_ninf);
        out[3 * (size_t)p + 2] = (uint8_t)film_level(b, s_thr, lv_nan, lv_neg, lv_ninf);
    }
}

} // namespace

namespace rt {
hipError_t launch_film(const float *fb_rgb, uint8_t *out_rgb8, uint32_t n_pixels, uint32_t shard_index, uint32_t shard_count, uint32_t shard_block,
                       const FilmTable *d_table, hipStream_t stream) {
    if (n_pixels == 0)
        return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n_pixels + 255) / 256, 256u * 16u);
    (void)hipGetLastError(); // clear the thread's sticky error state: what is returned below belongs to this launch
    hipLaunchKernelGGL(film_kernel, dim3(blocks), dim3(256), 0, stream, fb_rgb, out_rgb8, n_pixels, shard_index, shard_count ? shard_count : 1u,
                       shard_block ? shard_block : n_pixels, d_table);
    return hipGetLastError();
}
} // namespace rt

// valu_rates.hip — issue cost (cycles per wave-instruction) of a few VALU ops on gfx950, one wave per SIMD.
// Development aid: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 32768
#define REP8(x) x x x x x x x x
template <int OP> __global__ void k(float *out, unsigned long long *cyc, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT; ++i) {
        if (OP == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 1) { REP8(asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %2, %2, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 2) { REP8(asm volatile("v_cvt_f64_f32 %0, %1\n v_cvt_f64_f32 %2, %3" : "+v"(d0), "+v"(a1), "+v"(d2), "+v"(a3));) }
        if (OP == 3) { REP8(asm volatile("v_cvt_f32_f64 %0, %1\n v_cvt_f32_f64 %2, %3" : "+v"(a0), "+v"(d1), "+v"(a2), "+v"(d3));) }
        if (OP == 4) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %2, %2, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 5) { REP8(asm volatile("v_rcp_f32 %0, %1\n v_rcp_f32 %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 6) { REP8(asm volatile("v_max_f32 %0, %0, %1\n v_min_f32 %2, %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 7) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc");) }
        if (OP == 12) { REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 13) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 14) { REP8(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %2, %2, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc");) }
        if (OP == 15) { REP8(asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1\n v_cmp_lt_f32_e64 s[12:13], %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"s10","s11","s12","s13");) }
        if (OP == 8) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %2, %2, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 9) { REP8(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %2, %2, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 10) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 11) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %3, %0\n v_mad_u64_u32 %2, vcc, %1, %3, %2" : "+v"(d0), "+v"(a1), "+v"(d2), "+v"(a3)::"vcc");) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3);
}
template <int OP> void run(const char *name, float *out, unsigned long long *cyc, int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256, 256 * waves_per_simd>>>(out, cyc, 1.0f); // one block per CU, 4*w waves
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<256, 256 * waves_per_simd>>>(out, cyc, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n_instr = (double)N_IT * 16;
    // s_memtime/readcyclecounter ticks at a constant 100 MHz on gfx9; use wall clock with an assumed 2.4 GHz instead
    printf("%-16s w=%d  %.3f ms  -> %.2f cycles/instr/wave @2.4GHz (counter ticks %llu)\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / (n_instr * waves_per_simd), c);
}
int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    for (int w = 2; w <= 4; w *= 2) {
        run<0>("v_fma_f32", out, cyc, w); run<1>("v_mul_f64", out, cyc, w); run<2>("v_cvt_f64_f32", out, cyc, w); run<3>("v_cvt_f32_f64", out, cyc, w);
        run<4>("v_fma_f64", out, cyc, w); run<5>("v_rcp_f32", out, cyc, w); run<6>("v_max/min_f32", out, cyc, w); run<7>("v_cndmask_b32", out, cyc, w);
        run<8>("v_pk_fma_f32", out, cyc, w); run<9>("v_add_f64", out, cyc, w); run<10>("v_mul_lo_u32", out, cyc, w); run<12>("cndmask_e32 vcc", out, cyc, w); run<13>("cndmask_e64 sgpr", out, cyc, w); run<14>("cmp+cndmask", out, cyc, w); run<15>("v_cmp_e64 x2", out, cyc, w); run<11>("v_mad_u64_u32", out, cyc, w);
    }
    return 0;
}

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
        if (OP == 16) { REP8(asm volatile("v_cvt_f32_ubyte0 %0, %1\n v_cvt_f32_ubyte2 %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 17) { REP8(asm volatile("v_max3_f32 %0, %0, %1, %1\n v_min3_f32 %2, %2, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 18) { REP8(asm volatile("v_and_b32 %0, %0, %1\n v_lshlrev_b32 %2, 3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 19) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 20) { REP8(asm volatile("v_bfe_u32 %0, %1, 8, 20\n v_bcnt_u32_b32 %2, %3, %2" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 21) { REP8(asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %2, %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 22) { REP8(asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32_e64 %2, 0, 1, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc");) }
        // what the compiler emits for the wide node step's near / far plane selection: ONE compare, then several selects on the same vcc
        if (OP == 23) { REP8(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %3, %3, %2, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc");) }
        if (OP == 24) { REP8(asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n v_cndmask_b32_e64 %3, %3, %2, s[10:11]\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"s10","s11");) }
        if (OP == 25) { REP8(asm volatile("s_mov_b64 vcc, s[10:11]\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %1, %1, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc");) }
        if (OP == 26) { REP8(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %1, %1, %0, %0\n v_cndmask_b32_e32 %2, %2, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc");) }
        if (OP == 27) { REP8(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_fma_f32 %0, %0, %1, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc");) }
        if (OP == 28) { REP8(asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %1, %1, %0, %0\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"s10","s11");) }
        if (OP == 29) { REP8(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_fma_f32 %0, %0, %1, %1\n v_cndmask_b32_e32 %3, %3, %2, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"vcc");) }
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
    for (int w = 4; w <= 4; w *= 2) {
        run<0>("v_fma_f32", out, cyc, w); run<1>("v_mul_f64", out, cyc, w); run<2>("v_cvt_f64_f32", out, cyc, w); run<3>("v_cvt_f32_f64", out, cyc, w);
        run<4>("v_fma_f64", out, cyc, w); run<5>("v_rcp_f32", out, cyc, w); run<6>("v_max/min_f32", out, cyc, w); run<7>("v_cndmask_b32", out, cyc, w);
        run<8>("v_pk_fma_f32", out, cyc, w); run<9>("v_add_f64", out, cyc, w); run<10>("v_mul_lo_u32", out, cyc, w); run<12>("cndmask_e32 vcc", out, cyc, w); run<13>("cndmask_e64 sgpr", out, cyc, w); run<14>("cmp+cndmask", out, cyc, w); run<15>("v_cmp_e64 x2", out, cyc, w); run<11>("v_mad_u64_u32", out, cyc, w);
        run<16>("v_cvt_f32_ubyteN", out, cyc, w); run<17>("v_max3/min3_f32", out, cyc, w); run<18>("v_and/lshlrev", out, cyc, w); run<19>("v_mov_b32", out, cyc, w); run<20>("v_bfe/bcnt", out, cyc, w);
        run<21>("v_sub/mul_f32", out, cyc, w); run<22>("cmp + cndmask(imm)", out, cyc, w);
        run<23>("cmp, 3 cndmask vcc (x2 instr)", out, cyc, w); run<24>("cmp, 3 cndmask sgpr (x2)", out, cyc, w); run<25>("s_mov vcc, 3 cndmask (x2)", out, cyc, w);
        run<26>("cmp, 2 fma, cndmask vcc (x2)", out, cyc, w); run<27>("cmp, s_nop 1, cndmask vcc, fma (x2)", out, cyc, w); run<28>("cmp, 2 fma, cndmask sgpr (x2)", out, cyc, w);
        run<29>("cmp, cndmask, fma, cndmask vcc (x2)", out, cyc, w);
    }
    return 0;
}

/*
 * rt_devspec.h — the parts of the sample loop whose arithmetic is DEFINED by this project rather than by the
 * reference, written once so that the HIP kernels and the CPU oracle (oracle/rt_oracle.cpp, device-RNG mode)
 * evaluate the very same IEEE-754 operation sequence:
 *
 *   1. rt_xoshiro  : xoshiro128++ (Blackman & Vigna 2019) per-lane generator, state derived by splitmix64 from
 *                    (seed, pixel index, sample index). Replaces the reference's std::minstd_rand stream
 *                    (raytracer.h:458,648) in RT_RNG_DEVICE mode; draw ORDER stays the reference's.
 *   2. rt_minstd   : std::minstd_rand + the libstdc++-11 uniform_real<float> / uniform_int<int> algorithms
 *                    (bits/random.tcc generate_canonical; bits/uniform_int_dist.h:277-330, the two-division
 *                    "fallback" downscaling branch), used in RT_RNG_REFERENCE mode.
 *   3. rt_sincos_libm : glibc's sinf / cosf restated bit for bit, which is what the reference's std::sin / std::cos
 *                    (raytracer.h:104,158-159) evaluate; used by the device in BOTH RNG modes (the oracle calls libm itself).
 *   4. rt_atan2f_libm / rt_asinf_libm / rt_bg_uv : glibc's atan2f / asinf, for Scene::bg_at (scene.h:83-89).
 *
 * Only +,-,*,/ on float/double and integer ops are used; with floating-point contraction disabled
 * (-ffp-contract=off, both compilers) the results are bit-identical on x86-64 and gfx950.
 */
#ifndef RT_DEVSPEC_H
#define RT_DEVSPEC_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RT_HD __host__ __device__ inline
#else
#define RT_HD static inline
#endif

/* ---------------------------------------------------------------- xoshiro128++ ---- */
typedef struct rt_xoshiro {
    uint32_t s[4];
} rt_xoshiro;

RT_HD uint64_t rt_splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

RT_HD void rt_xoshiro_seed(rt_xoshiro *g, uint64_t seed, uint32_t pixel, uint32_t sample) {
    uint64_t x = seed ^ (((uint64_t)pixel << 32) | (uint64_t)sample) * 0xD1342543DE82EF95ull;
    uint64_t a = rt_splitmix64(&x);
    uint64_t b = rt_splitmix64(&x);
    g->s[0] = (uint32_t)a;
    g->s[1] = (uint32_t)(a >> 32);
    g->s[2] = (uint32_t)b;
    g->s[3] = (uint32_t)(b >> 32);
    if ((g->s[0] | g->s[1] | g->s[2] | g->s[3]) == 0u)
        g->s[0] = 1u;
}

RT_HD uint32_t rt_rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

RT_HD uint32_t rt_xoshiro_next(rt_xoshiro *g) {
    uint32_t r = rt_rotl32(g->s[0] + g->s[3], 7) + g->s[0];
    uint32_t t = g->s[1] << 9;
    g->s[2] ^= g->s[0];
    g->s[3] ^= g->s[1];
    g->s[1] ^= g->s[2];
    g->s[0] ^= g->s[3];
    g->s[2] ^= t;
    g->s[3] = rt_rotl32(g->s[3], 11);
    return r;
}

/* canonical float in [0,1): top 24 bits, exact */
RT_HD float rt_xoshiro_canonical(rt_xoshiro *g) {
    return (float)(rt_xoshiro_next(g) >> 8) * 5.9604644775390625e-08f; /* 2^-24 */
}

/* integer in [0, n): multiply-shift (deterministic; the tiny bias is irrelevant here) */
RT_HD uint32_t rt_xoshiro_below(rt_xoshiro *g, uint32_t n) {
    return (uint32_t)(((uint64_t)rt_xoshiro_next(g) * (uint64_t)n) >> 32);
}

/* ---------------------------------------------------------------- minstd_rand ---- */
typedef struct rt_minstd {
    uint32_t x;
} rt_minstd;

/* std::linear_congruential_engine<uint_fast32_t,48271,0,2147483647>::seed (bits/random.tcc): a seed that is
 * 0 mod m becomes 1 — hence spans 0 and 1 share a stream. */
RT_HD void rt_minstd_seed(rt_minstd *g, uint32_t seed) {
    uint32_t v = seed % 2147483647u;
    g->x = v == 0u ? 1u : v;
}

RT_HD uint32_t rt_minstd_next(rt_minstd *g) {
    g->x = (uint32_t)(((uint64_t)g->x * 48271ull) % 2147483647ull);
    return g->x;
}

/* generate_canonical<float,24>(minstd): m = 1 round; float(x - 1) / float(2147483646.0L) where the divisor
 * rounds to 2^31; results >= 1 are replaced by nextafter(1,0). */
RT_HD float rt_minstd_canonical(rt_minstd *g) {
    float s = (float)(rt_minstd_next(g) - 1u);
    float r = s / 2147483648.0f;
    if (r >= 1.0f)
        r = 0.99999994039535522461f;
    return r;
}

/* uniform_int_distribution<int>(0, n-1)(minstd), n >= 1 */
RT_HD uint32_t rt_minstd_below(rt_minstd *g, uint32_t n) {
    const uint64_t urngrange = 2147483645ull; /* max - min */
    const uint64_t uerange = (uint64_t)n;     /* urange + 1 */
    const uint64_t scaling = urngrange / uerange;
    const uint64_t past = uerange * scaling;
    uint64_t ret;
    do {
        ret = (uint64_t)rt_minstd_next(g) - 1ull;
    } while (ret >= past);
    return (uint32_t)(ret / scaling);
}

/* ---------------------------------------------------------------- sin / cos ---- */
/* The reference calls std::sin / std::cos on floats (raytracer.h:104,158-159): glibc's sinf / cosf. In RT_RNG_REFERENCE mode the
 * device evaluates THAT function: glibc 2.35 sysdeps/ieee754/flt-32/{s_sinf.c, s_cosf.c, sincosf.h} (the ARM Optimized Routines
 * algorithm: double-precision arithmetic, reduction by multiples of pi/2 with a 2^24-scaled 2/pi, two polynomials), restated with
 * its constants (they are also what the installed libm.so.6 holds in __sincosf_table). Only the branch structure for
 * 0 <= y < 120 is restated; the render loop's arguments are azimuths in [0, 2*pi]. tools/proofs/sincosf_exhaustive.c compares this
 * function with the host libm on EVERY float in [0, 2*pi] (1 086 918 636 values, both with and without contraction of the
 * multiply-adds: the final rounding to float absorbs the difference) — run by the CPU suite. With it the GPU's reference-RNG
 * render is the reference binary's image byte for byte, not merely the oracle's. */
/* sinf_poly's two branches: the sine polynomial in x (sign applied by the caller) and the cosine polynomial, whose coefficients the second
 * table entry negates (cs = -1) */
RT_HD float rt_libm_sin_poly(double x, double x2) {
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const double x3 = x * x2;
    const double s1 = S2 + x2 * S3;
    const double x7 = x3 * x2;
    const double s = x + x3 * S1;
    return (float)(s + x7 * s1);
}
RT_HD float rt_libm_cos_poly(double x2, double cs) {
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double x4 = x2 * x2;
    const double c2 = cs * C3 + x2 * (cs * C4);
    const double c1 = cs * C0 + x2 * (cs * C1);
    const double x6 = x4 * x2;
    const double c = c1 + x4 * (cs * C2);
    return (float)(c + x6 * c2);
}
RT_HD void rt_sincos_libm(float y, float *s_out, float *c_out) {
    const double HPI_INV = 0x1.45f306dc9c883p+23, HPI = 0x1.921fb54442d18p+0; /* 2^24 * 2/pi, pi/2 */
    union {
        float f;
        uint32_t u;
    } b;
    b.f = y;
    const uint32_t top12 = (b.u >> 20) & 0x7ffu; /* abstop12 */
    double x = (double)y;
    /* reduce_fast: n = round(x * 2/pi) through the 2^24-scaled product, x -= n * pi/2. glibc skips it below pi/4-ish (|y| < 0.75), where it
     * yields n = 0 and leaves x alone: one straight-line path serves both (no divergence between the lanes of a wave). */
    const double r = x * HPI_INV;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * HPI;
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; /* sign[n & 3] = {1, -1, -1, 1} */
    const double cs = (n & 2) ? -1.0 : 1.0;
    /* sinf_poly (x * sgn, x * x, table entry, n) gives the sine for even n and the cosine for odd n; cosf asks for n + 1: each polynomial once */
    const float ps = rt_libm_sin_poly(x * sgn, x * x), pc = rt_libm_cos_poly(x * x, cs);
    const int even = (n & 1) == 0;
    const int tiny = top12 < 0x398u; /* |y| < 2^-12: sinf returns y, cosf returns 1 */
    *s_out = tiny ? y : (even ? ps : pc);
    *c_out = tiny ? 1.0f : (even ? pc : ps);
}

/* ---------------------------------------------------------------- atan2 / asin, the reference's own ---- */
/* Scene::bg_at (scene.h:83-89) maps a direction to environment-map coordinates with std::atan2 / std::asin on floats: glibc's atan2f /
 * asinf, which on x86-64 are the single-precision fdlibm routines (sysdeps/ieee754/flt-32/{e_atan2f.c, s_atanf.c, e_asinf.c}: float
 * arithmetic only, no FMA variant is selected at run time). Restated here with their constants and compared with the installed libm by
 * tools/proofs/atan2f_asinf_exhaustive.c: atanf and asinf on ALL 2^32 floats, atan2f on 10^9 random pairs (uniform bit patterns and pairs
 * of comparable magnitude) plus the special-case grid — 0 mismatches. The errno-setting wrappers do not change values. Used in BOTH RNG
 * modes: the environment lookup is the reference's arithmetic, not a definition of ours. */
RT_HD uint32_t rt_f2u(float f) {
    union {
        float f;
        uint32_t u;
    } b;
    b.f = f;
    return b.u;
}
RT_HD float rt_u2f(uint32_t u) {
    union {
        float f;
        uint32_t u;
    } b;
    b.u = u;
    return b.f;
}
RT_HD float rt_atanf_libm(float x) {
    const float hi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f}; /* atan(0.5), atan(1), atan(1.5), atan(inf) */
    const float lo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float a0 = 3.3333334327e-01f, a1 = -2.0000000298e-01f, a2 = 1.4285714924e-01f, a3 = -1.1111110449e-01f, a4 = 9.0908870101e-02f,
                a5 = -7.6918758452e-02f, a6 = 6.6610731184e-02f, a7 = -5.8335702866e-02f, a8 = 4.9768779427e-02f, a9 = -3.6531571299e-02f,
                a10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)rt_f2u(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) { /* |x| >= 2^25 */
        if (ix > 0x7f800000)
            return x + x;
        return hx > 0 ? hi[3] + lo[3] : -hi[3] - lo[3];
    }
    if (ix < 0x3ee00000) { /* |x| < 0.4375 */
        if (ix < 0x31000000)
            return x; /* |x| < 2^-29 */
        id = -1;
    } else {
        x = rt_u2f((uint32_t)ix);
        if (ix < 0x3f980000) { /* |x| < 1.1875 */
            if (ix < 0x3f300000) {
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else if (ix < 0x401c0000) { /* |x| < 2.4375 */
            id = 2;
            x = (x - 1.5f) / (1.0f + 1.5f * x);
        } else {
            id = 3;
            x = -1.0f / x;
        }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0)
        return x - x * (s1 + s2);
    z = hi[id] - ((x * (s1 + s2) - lo[id]) - x);
    return hx < 0 ? -z : z;
}
RT_HD float rt_atan2f_libm(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)rt_f2u(x), hy = (int32_t)rt_f2u(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000)
        return x + y;
    if (hx == 0x3f800000)
        return rt_atanf_libm(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2); /* 2 * sign(x) + sign(y) */
    if (iy == 0)
        return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0)
        return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000)
            return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000)
        return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else
        z = rt_atanf_libm(rt_u2f(rt_f2u(y / x) & 0x7fffffffu));
    switch (m) {
    case 0:
        return z;
    case 1:
        return rt_u2f(rt_f2u(z) ^ 0x80000000u);
    case 2:
        return pi - (z - pi_lo);
    default:
        return (z - pi_lo) - pi;
    }
}
RT_HD float rt_asinf_libm(float x) {
    const float pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f, pio4_hi = 0.785398185253143310546875f;
    const float p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f, p4 = 4.216630880e-2f;
    const int32_t hx = (int32_t)rt_f2u(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000)
        return x * pio2_hi + x * pio2_lo;
    if (ix > 0x3f800000)
        return (x - x) / (x - x); /* |x| > 1 or NaN: NaN */
    if (ix < 0x3f000000) { /* |x| < 0.5 */
        if (ix < 0x32000000)
            return x;
        const float t = x * x;
        const float w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
        return x + x * w;
    }
    float w = 1.0f - rt_u2f((uint32_t)ix);
    float t = w * 0.5f;
    float p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    const float s = __builtin_sqrtf(t);
    if (ix >= 0x3F79999A) { /* |x| > 0.975 */
        t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
    } else {
        w = rt_u2f(rt_f2u(s) & 0xfffff000u);
        const float c = (t - w * w) / (s + w);
        const float r = p;
        p = 2.0f * s * r - (pio2_lo - 2.0f * c);
        const float q = pio4_hi - 2.0f * w;
        t = pio4_hi - (p - q);
    }
    return hx > 0 ? t : -t;
}
/* Scene::bg_at's coordinates (scene.h:85-87): x = 0.5 + 0.5 * atan2(z, x) / pi_f evaluated in double (the 0.5 literals are doubles, the
 * float results are promoted) and rounded to float once; y = 0.5 - asin(y) / pi_f with the quotient in float and the difference in double. */
RT_HD void rt_bg_uv(float dx, float dy, float dz, float *u, float *v) {
    const float pi_f = 3.14159274101257324f; /* std::numbers::pi_v<float> */
    *u = (float)(0.5 + 0.5 * (double)rt_atan2f_libm(dz, dx) / (double)pi_f);
    *v = (float)(0.5 - (double)(rt_asinf_libm(dy) / pi_f));
}

#endif /* RT_DEVSPEC_H */

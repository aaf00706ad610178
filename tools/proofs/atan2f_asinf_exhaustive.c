// tools/proofs/atan2f_asinf_exhaustive.c — rt_atanf_libm / rt_asinf_libm / rt_atan2f_libm (include/rt_devspec.h: glibc's single-precision
// fdlibm routines restated) against the host libm: atanf and asinf on ALL 2^32 float bit patterns, atan2f on 2^30 pseudo-random pairs (half of
// them uniform bit patterns, half with y and x of comparable magnitude, where the quotient is interesting) and on the full grid of special
// operands. The device evaluates these where the reference's Scene::bg_at (scene.h:83-89) calls std::atan2 / std::asin. Exit code 0 = bit-identical
// everywhere (NaN results compared as NaN). Run by tests/test_host_and_abi.py (about 25 s on 8 threads).
//   gcc -O2 -ffp-contract=off -I include tools/proofs/atan2f_asinf_exhaustive.c -o atan2f_asinf_exhaustive -lm -lpthread
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>

#include "rt_devspec.h"

typedef struct {
    uint64_t lo, hi;
    unsigned long long bad_atan, bad_asin, bad_atan2, pairs;
} job;
static int same(float a, float b) { return rt_f2u(a) == rt_f2u(b) || (a != a && b != b); }
static uint64_t mix(uint64_t *s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static void *run(void *a) {
    job *j = (job *)a;
    uint64_t st = j->lo * 77ull + 5;
    for (uint64_t u = j->lo; u < j->hi; ++u) {
        const float x = rt_u2f((uint32_t)u);
        j->bad_atan += !same(rt_atanf_libm(x), atanf(x));
        j->bad_asin += !same(rt_asinf_libm(x), asinf(x));
        if ((u & 3) == 0) {
            const uint64_t r = mix(&st);
            const float yy = rt_u2f((uint32_t)r);
            float xx = rt_u2f((uint32_t)(r >> 32));
            if (u & 4)
                xx = rt_u2f((rt_f2u(yy) & 0x7f800000u) | ((uint32_t)(r >> 32) & 0x807fffffu));
            j->bad_atan2 += !same(rt_atan2f_libm(yy, xx), atan2f(yy, xx));
            j->pairs++;
        }
    }
    return 0;
}
int main(void) {
    enum { T = 8 };
    pthread_t th[T];
    job jb[T];
    const uint64_t N = 1ull << 32;
    for (int t = 0; t < T; ++t) {
        jb[t] = (job){N * t / T, N * (t + 1) / T, 0, 0, 0, 0};
        pthread_create(&th[t], 0, run, &jb[t]);
    }
    unsigned long long a = 0, b = 0, c = 0, pairs = 0;
    for (int t = 0; t < T; ++t) {
        pthread_join(th[t], 0);
        a += jb[t].bad_atan, b += jb[t].bad_asin, c += jb[t].bad_atan2, pairs += jb[t].pairs;
    }
    /* special operands: zeros, ones, infinities, NaN, denormals, huge and tiny magnitudes, both signs, every pair */
    const float sp[] = {0.0f, 1.0f, 0.5f, 2.0f, 1e-45f, 1e-38f, 1e-30f, 1e-20f, 1e20f, 1e30f, 3.4e38f, INFINITY, NAN, 0.4375f, 0.6875f, 1.1875f, 2.4375f, 33554432.0f};
    const int ns = (int)(sizeof sp / sizeof sp[0]);
    unsigned long long d = 0;
    for (int i = 0; i < 2 * ns; ++i)
        for (int k = 0; k < 2 * ns; ++k) {
            const float y = i < ns ? sp[i] : -sp[i - ns], x = k < ns ? sp[k] : -sp[k - ns];
            d += !same(rt_atan2f_libm(y, x), atan2f(y, x));
        }
    printf("atanf: 4294967296 floats, mismatches %llu; asinf: 4294967296 floats, mismatches %llu; atan2f: %llu random pairs, mismatches %llu; %d special pairs, mismatches %llu\n",
           a, b, pairs, c, 4 * ns * ns, d);
    return (a | b | c | d) != 0;
}

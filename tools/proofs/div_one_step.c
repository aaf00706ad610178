/* div_one_step.c — exhaustive hard-case check of the one-step exact division used by the slab test
 * (csrc/rt_device_lib.h div_exact_fast):
 *
 *     r  = RN(1/d)            (once per ray, IEEE division)
 *     q0 = RN(a*r);  e = RN(a - d*q0) (one FMA);  q1 = RN(q0 + e*r) (one FMA);   claim: q1 == RN(a/d)
 *
 * for all binary32 a, d whose quotient and intermediate values stay in the normal range (the per-ray / per-scene
 * preconditions of the fast path). Argument: with r = (1/d)(1+eps), |eps| <= 2^-24, the value rounded in the last step
 * is v = Q + (Q - q0)*eps (+ the rounding of e when e is not exact), Q = a/d, so |v - Q| <= ~3*2^-24 ulp(Q). q1 can
 * differ from RN(Q) only if a rounding boundary (midpoint m of two neighbouring floats) lies between v and Q. For
 * significands A, B in [2^23, 2^24) the distance of A/B to a midpoint m/2^k (m odd) is |A*2^k - B*m| / (B*2^k) with
 * k = 24 for A >= B (Q in [1,2)) and k = 25 for A < B (Q in (1/2,1)); in ulps that is |c| / (2B) with the non-zero integer
 * c = A*2^k - B*m. A failure therefore needs |c| < 8. This program enumerates EVERY (A, B, m) with 0 < |c| <= 16 by
 * solving B*m = -c (mod 2^k) for each B, runs the float sequence on each pair and compares it with IEEE division
 * (plus a random sweep as a sanity check of the harness). The operations are invariant under scaling by powers of two
 * inside the normal range, so significand pairs cover all exponents.
 * Build/run: gcc -O2 div_one_step.c -o div_one_step -lm && ./div_one_step     (prints "failures 0", exit code 0)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CMAX 16

static uint64_t q0_differs = 0; /* harness self-check: the uncorrected product must be wrong on many hard cases */
static int check(uint32_t A, uint32_t B, int two_step) {
    volatile float a = (float)A, d = (float)B; /* exact: 24-bit integers */
    volatile float q = a / d;
    float r = 1.0f / d;
    float q0 = a * r;
    float e0 = fmaf(-d, q0, a);
    float q1 = fmaf(e0, r, q0);
    q0_differs += (q0 != q);
    if (two_step) {
        float e1 = fmaf(-d, q1, a);
        q1 = fmaf(e1, r, q1);
    }
    return q1 == q;
}

static uint64_t inv_odd(uint64_t b, int bits) { /* inverse of odd b modulo 2^bits (Newton) */
    uint64_t x = b; /* correct to 3 bits */
    for (int i = 0; i < 6; ++i)
        x *= 2 - b * x;
    return bits >= 64 ? x : (x & ((1ull << bits) - 1));
}

int main(void) {
    uint64_t tested = 0, failures = 0, failures2 = 0;
    for (uint32_t B = 1u << 23; B < (1u << 24); ++B) {
        const int tz = __builtin_ctz(B);
        const int64_t g = 1ll << tz;
        if (g > CMAX)
            continue;
        const uint64_t Bo = B >> tz;
        for (int k = 24; k <= 25; ++k) {
            const int mb = k - tz;
            const uint64_t mod = 1ull << mb, inv = inv_odd(Bo, mb);
            for (int64_t co = -(CMAX / g) | 1; co * g <= CMAX; co += 2) { /* odd multipliers, c = co * g */
                if (co * g < -CMAX)
                    continue;
                const int64_t c = co * g;
                const uint64_t m_base = ((uint64_t)(-co) * inv) & (mod - 1);
                for (uint64_t j = 0; j < (1ull << tz); ++j) {
                    uint64_t m = m_base + j * mod; /* residue modulo 2^k */
                    if (k == 24)
                        m += 1ull << 24; /* the representative in [2^24, 2^25) */
                    if (m < (1ull << 24) || m >= (1ull << 25) || !(m & 1))
                        continue;
                    const int64_t num = (int64_t)((uint64_t)B * m) + c;
                    if (num <= 0 || (num & ((1ll << k) - 1)))
                        continue;
                    const uint64_t A = (uint64_t)num >> k;
                    if (A < (1u << 23) || A >= (1u << 24))
                        continue;
                    if ((k == 24) != (A >= B))
                        continue;
                    ++tested;
                    if (!check((uint32_t)A, B, 0)) {
                        if (failures < 10)
                            printf("FAIL one-step A=%u B=%u c=%lld\n", (unsigned)A, B, (long long)c);
                        ++failures;
                    }
                    failures2 += !check((uint32_t)A, B, 1);
                }
            }
        }
    }
    uint64_t s = 88172645463325252ull, rnd_fail = 0;
    for (int i = 0; i < 200000000; ++i) { /* harness sanity: random significand pairs */
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        const uint32_t A = (1u << 23) | ((uint32_t)s & 0x7FFFFF), B = (1u << 23) | ((uint32_t)(s >> 32) & 0x7FFFFF);
        rnd_fail += !check(A, B, 0);
    }
    printf("hard cases tested %llu, failures %llu (two-step sequence: %llu), random failures %llu, uncorrected a*r wrong in %llu checks\n",
           (unsigned long long)tested, (unsigned long long)failures, (unsigned long long)failures2, (unsigned long long)rnd_fail,
           (unsigned long long)q0_differs);
    return failures == 0 && rnd_fail == 0 && tested > 1000000 && q0_differs > 1000000 ? 0 : 1;
}

// tools/proofs/sincosf_exhaustive.c — rt_sincos_libm (include/rt_devspec.h: glibc 2.35's sinf / cosf restated) against the host libm on
// EVERY float in [0, 2*pi]: the device's reference-RNG mode evaluates this function where the reference calls std::sin / std::cos
// (raytracer.h:104,158-159). Exit code 0 = bit-identical on all of them. Run by tests/test_host_and_abi.py (a few seconds on 8 threads).
//   gcc -O2 -ffp-contract=off -I include tools/proofs/sincosf_exhaustive.c -o sincosf_exhaustive -lm -lpthread
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "rt_devspec.h"

typedef struct {
    uint32_t lo, hi, first;
    unsigned long long bad;
} job;
static void *run(void *a) {
    job *j = (job *)a;
    for (uint32_t u = j->lo; u < j->hi; ++u) {
        float y, s, c;
        memcpy(&y, &u, 4);
        rt_sincos_libm(y, &s, &c);
        const float ls = sinf(y), lc = cosf(y);
        if (memcmp(&s, &ls, 4) || memcmp(&c, &lc, 4)) {
            if (!j->bad)
                j->first = u;
            j->bad++;
        }
    }
    return 0;
}
int main(void) {
    const float top = 6.2831860f; /* the first float above 2*pi */
    uint32_t hi;
    memcpy(&hi, &top, 4);
    hi += 16;
    enum { T = 8 };
    pthread_t th[T];
    job jobs[T];
    for (int t = 0; t < T; ++t) {
        jobs[t] = (job){(uint32_t)((uint64_t)hi * t / T), (uint32_t)((uint64_t)hi * (t + 1) / T), 0, 0};
        pthread_create(&th[t], 0, run, &jobs[t]);
    }
    unsigned long long bad = 0;
    uint32_t first = 0;
    for (int t = 0; t < T; ++t) {
        pthread_join(th[t], 0);
        if (!bad && jobs[t].bad)
            first = jobs[t].first;
        bad += jobs[t].bad;
    }
    printf("floats checked: %u, mismatches against libm sinf/cosf: %llu (first at 0x%08x)\n", hi, bad, first);
    return bad != 0;
}

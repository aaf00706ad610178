// rt_device_lib.h — device-side building blocks shared by the render kernels (rt_kernels.hip: persistent megakernel
// + probe kernels; rt_wavefront.hip: the wavefront pipeline): float3 algebra with the reference's operation order,
// exact slab / triangle tests, the resumable BVH traversal step, the light-pdf traversal, texture sampling, the shading
// record, sampling distributions and the BRDF. Reference citations are next to each function.
//
// Arithmetic contract: IEEE binary32, correctly rounded / and sqrt, no FMA contraction (-ffp-contract=off); FMA only
// where written explicitly (div_exact_fast). std::min/std::max operand order is reproduced by explicit selects.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/rt_abi.h"
#include "../../include/rt_devspec.h"
#include "../../include/rt_primspec.h"
#include "rt_device_types.h"

namespace {

constexpr float EPS = 1e-4;               // config.h:15
constexpr float MIN_ROUGHNESS = 0.04f;    // config.h:20
constexpr float VNDF_FACTOR = 1.0f / 3;   // config.h:26
constexpr float PI_F = 3.14159265358979323846f;
#define RT_INF __builtin_inff()
#define RT_NAN __builtin_nanf("")

#define DEV __device__ __forceinline__

struct V3 {
    float x, y, z;
};
DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
DEV V3 ld3(const float *p) { return V3{p[0], p[1], p[2]}; }
DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
DEV V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
DEV V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
DEV V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
DEV V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
DEV V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
DEV V3 operator-(float s, V3 a) { return {s - a.x, s - a.y, s - a.z}; }
DEV V3 operator-(V3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
DEV float len(V3 a) { return __builtin_sqrtf(len2(a)); }
DEV V3 crs(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; } // geometry.h:18-24
DEV V3 norm(V3 v) { return v / len(v); }                                                                   // geometry.h:31-34
DEV float rmin(float a, float b) { return (b < a) ? b : a; } // std::min(a,b)
DEV float rmax(float a, float b) { return (a < b) ? b : a; } // std::max(a,b)
DEV V3 transform3(V3 l, V3 x, V3 y, V3 z) { return l.x * x + l.y * y + l.z * z; } // geometry.h:355-359
DEV float pow2(float x) { return x * x; }
DEV float pow5(float x) { // raytracer.h:28-38, p = 5
    float x2 = x * x;
    return x * ((x2 * x2) * 1.0f);
}
DEV bool isnan_f(float x) { return x != x; }

struct C4 {
    float r, g, b, a;
};
DEV C4 operator*(float s, C4 c) { return {s * c.r, s * c.g, s * c.b, s * c.a}; }
DEV C4 operator+(C4 a, C4 b) { return {a.r + b.r, a.g + b.g, a.b + b.b, a.a + b.a}; }
DEV C4 operator*(C4 a, C4 b) { return {a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a}; }

// ---------------------------------------------------------------------------------------------- counters
template <bool ON> struct LaneStats;
template <> struct LaneStats<false> {
    DEV void cast() {}
    DEV void node() {}
    DEV void box(uint32_t) {}
    DEV void tri() {}
    DEV void shaded() {}
    DEV void lq() {}
    DEV void lnode() {}
    DEV void lbox(uint32_t) {}
    DEV void ltri() {}
    DEV void lhit() {}
    DEV void texels(uint32_t) {}
    DEV void sample() {}
    DEV void flush(DevStats *) {}
};
template <> struct LaneStats<true> {
    unsigned long long c_cast = 0, c_node = 0, c_box = 0, c_tri = 0, c_shaded = 0, c_lq = 0, c_lnode = 0, c_lbox = 0, c_ltri = 0, c_lhit = 0,
                       c_tex = 0, c_sample = 0;
    DEV void cast() { ++c_cast; }
    DEV void node() { ++c_node; }
    DEV void box(uint32_t n) { c_box += n; }
    DEV void tri() { ++c_tri; }
    DEV void shaded() { ++c_shaded; }
    DEV void lq() { ++c_lq; }
    DEV void lnode() { ++c_lnode; }
    DEV void lbox(uint32_t n) { c_lbox += n; }
    DEV void ltri() { ++c_ltri; }
    DEV void lhit() { ++c_lhit; }
    DEV void texels(uint32_t n) { c_tex += n; }
    DEV void sample() { ++c_sample; }
    DEV void flush(DevStats *s) {
        if (!s)
            return;
        atomicAdd(&s->casts, c_cast);
        atomicAdd(&s->nodes, c_node);
        atomicAdd(&s->box_tests, c_box);
        atomicAdd(&s->tri_tests, c_tri);
        atomicAdd(&s->shaded, c_shaded);
        atomicAdd(&s->lq, c_lq);
        atomicAdd(&s->lnodes, c_lnode);
        atomicAdd(&s->lbox, c_lbox);
        atomicAdd(&s->ltri, c_ltri);
        atomicAdd(&s->lhits, c_lhit);
        atomicAdd(&s->texels, c_tex);
        atomicAdd(&s->samples, c_sample);
    }
};

// Development-only wave-level execution census (-DRT_DIAG, tools_variants.sh): how often each section of the
// persistent loop runs and with how many active lanes. Reuses the DevStats words; never compiled into the product.
#ifdef RT_DIAG
__device__ DevStats *g_diag = nullptr;
DEV void diag_add(int slot, unsigned long long v) {
    const unsigned long long m = __ballot(1);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (rank == 0 && g_diag)
        atomicAdd(reinterpret_cast<unsigned long long *>(g_diag) + slot, v);
}
#define DIAG(slot, v) diag_add(slot, v)
#define DIAG_LANES(slot) diag_add(slot, (unsigned long long)__popcll(__ballot(1)))
#else
#define DIAG(slot, v) do { } while (0)
#define DIAG_LANES(slot) do { } while (0)
#endif

// Development-only section census of wf_shade (-DRT_DIAG_SHADE): wave cycles (s_memtime) and active lanes between stamps,
// accumulated per wave in LDS and flushed to the census words at kernel end. Never compiled into the product.
#ifdef RT_DIAG_SHADE
// A stamp at the END of a region (also inside a divergent branch or a loop body) attributes the wave cycles since the wave's previous stamp to
// `section`, once plain and once weighted with the lanes active at the stamp: lane_cycles / cycles = the lanes that section really runs at.
enum { SD_LOAD = 0, SD_ATTR, SD_TEX, SD_ALPHA, SD_S_VNDF, SD_S_COS, SD_S_LIGHT, SD_VNDF_PDF, SD_LPDF, SD_BRDF, SD_FOLD, SD_STORE, SD_N };
__shared__ unsigned long long g_sd_cyc[4][SD_N], g_sd_lanes[4][SD_N], g_sd_t[4];
DEV void sd_stamp(int section) {
    const unsigned long long m = __ballot(1);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    if (rank == 0) {
        const uint32_t w = threadIdx.x >> 6;
        const unsigned long long dt = now - g_sd_t[w];
        g_sd_cyc[w][section] += dt;
        g_sd_lanes[w][section] += dt * (unsigned long long)__popcll(m);
        g_sd_t[w] = now;
    }
}
#define SD_STAMP(section) sd_stamp(section)
#else
#define SD_STAMP(section) do { } while (0)
#endif


// ---------------------------------------------------------------------------------------------- RNG policy
template <int MODE> struct Rng;
template <> struct Rng<RT_RNG_DEVICE> {
    rt_xoshiro g;
    DEV float canonical() { return rt_xoshiro_canonical(&g); }
    DEV uint32_t below(uint32_t n) { return rt_xoshiro_below(&g, n); }
};
template <> struct Rng<RT_RNG_REFERENCE> {
    rt_minstd g;
    DEV float canonical() { return rt_minstd_canonical(&g); }
    DEV uint32_t below(uint32_t n) { return rt_minstd_below(&g, n); }
};
// std::uniform_real_distribution<float>(a, b)(rng) = canonical * (b - a) + a
template <class R> DEV float uniform_real(R &r, float a, float b) { return r.canonical() * (b - a) + a; }

// ---------------------------------------------------------------------------------------------- primitives
// Exact quotient a/d from a precomputed r = RN(1/d): q0 = RN(a*r), one FMA residual e = a - d*q0 and one FMA correction
// q1 = RN(q0 + e*r) = RN(a/d). 3 VALU ops instead of the ~11-op IEEE division expansion (v_div_scale / v_rcp / ... /
// v_div_fixup). Why one correction suffices although q0 may be off by 1.5 ulp (Markstein's theorem wants a faithful
// q0): the value rounded last is Q + (Q - q0)*eps with |eps| <= 2^-24, i.e. within ~3*2^-24 ulp of Q = a/d, and a
// quotient of two 24-bit significands comes that close to a rounding boundary only for the finitely many pairs with
// |A*2^k - B*m| <= 16 — all 93 million of which tools/proofs/div_one_step.c enumerates and checks against IEEE division
// (run by tests/test_host_and_abi.py). The residual must be exactly representable and nothing may overflow/underflow;
// that is guaranteed per RAY and per SCENE, not per box:
//   * every direction component has |d_i| in [2^-40, 2^40]                                   (trav_init)
//   * every origin component and every box coordinate is 0 or has magnitude in [2^-37, 2^40] (trav_init, host)
// so a = box - o is 0 or a multiple of 2^-60 with |a| <= 2^41, hence q = 0 or 2^-100 <= |q| <= 2^81, all normal.
// Rays (or scenes) outside these bounds take the reference IEEE division instead.
DEV float div_exact_fast(float a, float d, float r) {
    const float q0 = a * r;
    const float e0 = __builtin_fmaf(-d, q0, a);
    return __builtin_fmaf(e0, r, q0);
}
constexpr float RANGE_LO = 9.094947017729282e-13f;  // 2^-40
constexpr float RANGE_HI = 1099511627776.0f;        // 2^40
constexpr float ORIGIN_LO = 7.275957614183426e-12f; // 2^-37
DEV bool coord_in_fast_range(float c) { // 0, or 2^-37 <= |c| <= 2^40 (false for NaN / inf)
    const float m = __builtin_fabsf(c);
    return (c == 0.0f) | ((m >= ORIGIN_LO) & (m <= RANGE_HI));
}

// intersect(ray, aabb, min_dst) bvh.h:137-152, reference form: IEEE division, std::min/max operand order kept by
// explicit selects, component reductions as std::max_element / std::min_element (first extremum, geometry.h:42-50).
DEV bool box_hit_exact(V3 bmin, V3 bmax, V3 o, V3 d, float min_dst, float &dist) {
    V3 i1 = (bmin - o) / d;
    V3 i2 = (bmax - o) / d;
    V3 mn = {rmin(i1.x, i2.x), rmin(i1.y, i2.y), rmin(i1.z, i2.z)};
    V3 mx = {rmax(i1.x, i2.x), rmax(i1.y, i2.y), rmax(i1.z, i2.z)};
    float t_min = mn.x;
    if (t_min < mn.y)
        t_min = mn.y;
    if (t_min < mn.z)
        t_min = mn.z;
    float t_max = mx.x;
    if (mx.y < t_max)
        t_max = mx.y;
    if (mx.z < t_max)
        t_max = mx.z;
    if (t_min <= t_max && t_max >= min_dst) {
        dist = rmax(t_min, min_dst);
        return true;
    }
    return false;
}

// Same slab test on the fast path: the six quotients come from div_exact_fast and are the correctly rounded finite
// quotients (see above), so there is no NaN and no infinity among them and v_min/v_max agree with the reference's
// select forms up to the sign of a zero, which cannot reach the result: t_min/t_max are only compared, and
// max(t_min, min_dst) with min_dst = 1e-4 > 0 never returns a zero.
DEV bool box_hit_fast(V3 bmin, V3 bmax, V3 o, V3 d, V3 r, float min_dst, float &dist) {
    V3 a1 = bmin - o, a2 = bmax - o;
    float q1x = div_exact_fast(a1.x, d.x, r.x), q1y = div_exact_fast(a1.y, d.y, r.y), q1z = div_exact_fast(a1.z, d.z, r.z);
    float q2x = div_exact_fast(a2.x, d.x, r.x), q2y = div_exact_fast(a2.y, d.y, r.y), q2z = div_exact_fast(a2.z, d.z, r.z);
    float t_min = fmaxf(fmaxf(fminf(q1x, q2x), fminf(q1y, q2y)), fminf(q1z, q2z));
    float t_max = fminf(fminf(fmaxf(q1x, q2x), fmaxf(q1y, q2y)), fmaxf(q1z, q2z));
    dist = fmaxf(t_min, min_dst);
    return (t_min <= t_max) & (t_max >= min_dst);
}

// intersect_ray_triangle + intersect(ray, triangle, min_dst) bvh.h:36-65 (Cramer; xs = (b, c, t)).
// det(c1,c2,c3) = dot(c1, crs(c2,c3)) (geometry.h:26-29); crs(u, -d) is shared by two determinants.
// Division-free rejection filter. With D = |den| in [2^-60, 2^60] and sign-adjusted numerators n' = n * sign(den)
// (so the exact quotients are X = nx'/D, Y = ny'/D, Z = nz'/D) a triangle CERTAINLY fails the reference's test
//     xs.x >= 0 && xs.y >= 0 && xs.x + xs.y <= 1 && xs.z >= min_dst          (bvh.h:59-60, xs = RN(n/den))
// when  nx' < -2^-60  or  ny' < -2^-60           (X or Y < -2^-120: the rounded quotient is negative, not -0)
//   or  nx' + ny' > D * (1 + 2^-20)               (X + Y > 1 + 2^-20: beyond the three roundings, ~3 * 2^-24)
//   or  nz' < D * min_dst * (1 - 2^-20)           (Z < min_dst beyond the rounding of RN(Z))
// Anything else ("maybe") takes the reference's three IEEE divisions and its exact comparisons, so the filter only
// removes work, never changes an outcome. NaN/inf operands make every comparison false -> "maybe".
DEV bool tri_hit(V3 ta, V3 av, V3 au, V3 o, V3 d, float min_dst, V3 &xs_out) {
    V3 at = -d;
    V3 y = o - ta;
    V3 c_ut = crs(au, at);
    float den = dot(av, c_ut);
    float nx = dot(y, c_ut), ny = dot(av, crs(y, at)), nz = dot(av, crs(au, y));
    const uint32_t sgn = __float_as_uint(den) & 0x80000000u;
    const float D = __builtin_fabsf(den);
    const float nxs = __uint_as_float(__float_as_uint(nx) ^ sgn), nys = __uint_as_float(__float_as_uint(ny) ^ sgn),
                nzs = __uint_as_float(__float_as_uint(nz) ^ sgn);
    const bool d_ok = (D >= 8.673617379884035e-19f) & (D <= 1.152921504606847e18f); // 2^-60 .. 2^60
    const bool miss = (nxs < -8.673617379884035e-19f) | (nys < -8.673617379884035e-19f) | (nxs + nys > D * 1.00000095367431640625f) |
                      (nzs < D * (min_dst * 0.99999904632568359375f));
    if (d_ok & miss)
        return false;
    V3 xs = V3{nx, ny, nz} / den;
    if (xs.x >= 0 && xs.y >= 0 && xs.x + xs.y <= 1 && xs.z >= min_dst) {
        xs_out = xs;
        return true;
    }
    return false;
}

DEV DevTri load_tri(const DevTri *tris, uint32_t idx) {
    DevTri t;
    const float4 *p = reinterpret_cast<const float4 *>(tris + idx);
    float4 *q = reinterpret_cast<float4 *>(&t);
    q[0] = p[0];
    q[1] = p[1];
    q[2] = p[2];
    return t;
}

struct Hit {
    uint32_t k; // DevTri index (BVH order) or RT_NONE
    float b, c, t;
};

// Analytic primitives (scene-txt ELLIPSOID / PLANE, include/rt_primspec.h): tested by brute force AFTER the BVH, in index
// order, with the same strict-less replacement as update_intersection (bvh.h:132) — the CPU oracle does exactly this.
DEV rt_primitive_desc load_prim(const rt_primitive_desc *prims, uint32_t i) {
    rt_primitive_desc p;
    const uint4 *src = reinterpret_cast<const uint4 *>(prims + i); // 48-byte records, 16-byte aligned
    uint4 *dst = reinterpret_cast<uint4 *>(&p);
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
    return p;
}
DEV void prims_closest(const DevScene &S, V3 o, V3 d, Hit &best) {
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    for (uint32_t i = 0; i < S.n_prims; ++i) {
        const rt_primitive_desc p = load_prim(S.prims, i);
        float t, n[3];
        if (rt_prim_intersect(&p, oo, dd, 1e-4f /* EPS, config.h:15 */, &t, n) && (best.k == RT_NONE || best.t > t)) {
            best.k = RT_PRIM_FLAG | i;
            best.b = 0.0f;
            best.c = 0.0f;
            best.t = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------- traversal stack
// Deferred far siblings: {child ref, far entry distance, enclosing subtree's local best}. The first LDS_DEPTH
// positions live in LDS, one column per thread (bank = thread % 32: conflict free whatever the lanes' depths);
// deeper positions fall back to per-lane scratch. sp counts only ancestors whose BOTH children were hit, so the LDS
// part serves almost every access (DESIGN.md "traversal stack").
#ifndef RT_LDS_DEPTH
#define RT_LDS_DEPTH 12
#endif
constexpr int LDS_DEPTH = RT_LDS_DEPTH;
template <int LDS_DEPTH> struct StackMemT {
    uint32_t *lds; // [3][LDS_DEPTH][256] dwords, this thread's column starts at lds + threadIdx.x
    // The struct holds POINTERS only (registers after scalar replacement): with the arrays as members the whole object,
    // `lds` included, lived in scratch and every LDS access became a generic flat_load/flat_store behind a scratch load
    // of the pointer. RT_DECLARE_STACK sets the pointers up from a __shared__ array and per-lane overflow arrays.
    uint32_t *ov_ref; // [RT_MAX_STACK - LDS_DEPTH], per-lane scratch
    float *ov_d;
    float *ov_loc;
    DEV void push(int sp, uint32_t ref, float d, float loc) {
        if (sp < LDS_DEPTH) {
            lds[(0 * LDS_DEPTH + sp) * 256] = ref;
            lds[(1 * LDS_DEPTH + sp) * 256] = __float_as_uint(d);
            lds[(2 * LDS_DEPTH + sp) * 256] = __float_as_uint(loc);
        } else {
            ov_ref[sp - LDS_DEPTH] = ref;
            ov_d[sp - LDS_DEPTH] = d;
            ov_loc[sp - LDS_DEPTH] = loc;
        }
    }
    DEV void pop(int sp, uint32_t &ref, float &d, float &loc) {
        // The LDS read is unconditional (clamped slot) and the rare deep entry overrides it: selecting between an LDS and
        // a scratch POINTER would turn both into one generic flat_load.
        const int slot = sp < LDS_DEPTH ? sp : LDS_DEPTH - 1;
        ref = lds[(0 * LDS_DEPTH + slot) * 256];
        d = __uint_as_float(lds[(1 * LDS_DEPTH + slot) * 256]);
        loc = __uint_as_float(lds[(2 * LDS_DEPTH + slot) * 256]);
        if (sp >= LDS_DEPTH) {
            ref = ov_ref[sp - LDS_DEPTH];
            d = ov_d[sp - LDS_DEPTH];
            loc = ov_loc[sp - LDS_DEPTH];
        }
    }
    // pop() inside straight-line wave code: every lane reads some valid LDS slot (sp may be negative or stale on lanes
    // that do not `want` the frame), only wanting lanes look at the overflow part
    DEV void pop_masked(int sp, bool want, uint32_t &ref, float &d, float &loc) {
        const uint32_t slot = (uint32_t)sp < (uint32_t)LDS_DEPTH ? (uint32_t)sp : (uint32_t)(LDS_DEPTH - 1);
        ref = lds[(0 * LDS_DEPTH + slot) * 256];
        d = __uint_as_float(lds[(1 * LDS_DEPTH + slot) * 256]);
        loc = __uint_as_float(lds[(2 * LDS_DEPTH + slot) * 256]);
        // keep the LDS reads where they are: sunk into the branch below they would merge with the scratch loads into
        // generic flat_loads of a selected pointer
        asm volatile("" : "+v"(ref), "+v"(d), "+v"(loc));
        if (want && sp >= LDS_DEPTH) {
            ref = ov_ref[sp - LDS_DEPTH];
            d = ov_d[sp - LDS_DEPTH];
            loc = ov_loc[sp - LDS_DEPTH];
        }
    }
    // light-pdf traversal only needs child refs (bvh.h:237-260 has no ordering / pruning)
    DEV void push_ref(int sp, uint32_t ref) {
        if (sp < LDS_DEPTH)
            lds[(0 * LDS_DEPTH + sp) * 256] = ref;
        else
            ov_ref[sp - LDS_DEPTH] = ref;
    }
    DEV uint32_t pop_ref(int sp) {
        uint32_t ref = lds[(0 * LDS_DEPTH + (sp < LDS_DEPTH ? sp : LDS_DEPTH - 1)) * 256];
        if (sp >= LDS_DEPTH)
            ref = ov_ref[sp - LDS_DEPTH];
        return ref;
    }
};
// The same stack for wf_extend, where deep traversals are the norm (S-10M: a tree ~24 levels deep keeps 8-12 deferred
// siblings pending): the LDS part is a RING that always holds the NEWEST frames. Memory position p (0 = oldest) lives in
// LDS slot p % LDS_DEPTH while p >= base, and in scratch once it has been evicted (p < base). A push into a full ring evicts
// the OLDEST LDS frame to scratch; a pop that finds the ring empty takes one frame back from scratch. Scratch is touched only
// when the depth wanders further than LDS_DEPTH from where it was — with StackMemT's fixed split (positions >= LDS_DEPTH
// always in scratch) every push and pop beyond depth 7 went to scratch: 1.2 scratch pushes per cast on S-sponza (0.65 with
// the ring), 9x HBM write amplification of the kernel's real output, the hit records (profiles/r02_write_amp.txt).
template <int LDS_DEPTH, int WORDS = 3> struct RingStackT {
    // WORDS = 3: the reference traversal's frame {ref, d_far, saved local best}; WORDS = 2: the global-best traversal's {ref, d_far}
    static_assert(WORDS == 2 || WORDS == 3, "a frame is {ref, d_far} or {ref, d_far, saved local best}");
    using Rec = std::conditional_t<WORDS == 3, uint4, uint2>;
    uint32_t *lds; // [WORDS][LDS_DEPTH][256] dwords, this thread's column starts at lds + threadIdx.x
    // Evicted frames go to a global workspace, one record {ref, d_far[, saved local best, -]} per (position, thread),
    // laid out [position][thread of the grid]: an eviction or a refill is ONE 16-byte (8-byte) access (the three per-lane scratch
    // arrays this replaces cost three 4-byte accesses in three different 256-byte rows, i.e. three 32-byte sector writes
    // once the lines left L2), and neighbouring lanes' records of one position share lines.
    Rec *ov;        // this thread's record of position 0
    uint32_t stride; // records per position = threads of the grid
    int base;       // positions [base, newest] are in LDS
    DEV static uint32_t slot_of(uint32_t pos) { // pos % LDS_DEPTH for pos < 64
        if constexpr (LDS_DEPTH == 8)
            return pos & 7u;
        else if constexpr (LDS_DEPTH == 4)
            return pos & 3u;
        else if constexpr (LDS_DEPTH == 16)
            return pos & 15u;
        else {
            // pos / LDS_DEPTH by a full-rate 24-bit multiply (a 32-bit v_mul_lo_u32, which the compiler picks for a plain
            // `*` here, issues at quarter rate), then pos - LDS_DEPTH * q with shifts and adds
            uint32_t q;
            asm("v_mul_u32_u24 %0, %1, %2" : "=v"(q) : "v"(pos), "v"((uint32_t)((65536 + LDS_DEPTH - 1) / LDS_DEPTH)));
            q >>= 16;
            static_assert(LDS_DEPTH == 6 || LDS_DEPTH == 5 || LDS_DEPTH == 12 || LDS_DEPTH == 3 || LDS_DEPTH == 9 || LDS_DEPTH == 10 || LDS_DEPTH == 7, "add the shift/add form of LDS_DEPTH * q");
            const uint32_t m = LDS_DEPTH == 6    ? (q << 2) + (q << 1)
                               : LDS_DEPTH == 5  ? (q << 2) + q
                               : LDS_DEPTH == 12 ? (q << 3) + (q << 2)
                               : LDS_DEPTH == 9  ? (q << 3) + q
                               : LDS_DEPTH == 10 ? (q << 3) + (q << 1)
                               : LDS_DEPTH == 7  ? (q << 3) - q
                                                 : (q << 1) + q;
            return pos - m;
        }
    }
    DEV void reset() { base = 0; }
    DEV void push(int pos, uint32_t ref, float d, float loc = 0.0f) {
        const uint32_t slot = slot_of((uint32_t)pos);
        if (pos - base == LDS_DEPTH) { // ring full: the slot about to be overwritten holds position `base`, the oldest
            DIAG(28, (unsigned long long)__popcll(__ballot(1)));
            if constexpr (WORDS == 3)
                ov[(size_t)base * stride] = make_uint4(lds[(0 * LDS_DEPTH + slot) * 256], lds[(1 * LDS_DEPTH + slot) * 256], lds[(2 * LDS_DEPTH + slot) * 256], 0u);
            else
                ov[(size_t)base * stride] = make_uint2(lds[(0 * LDS_DEPTH + slot) * 256], lds[(1 * LDS_DEPTH + slot) * 256]);
            ++base;
        }
        lds[(0 * LDS_DEPTH + slot) * 256] = ref;
        lds[(1 * LDS_DEPTH + slot) * 256] = __float_as_uint(d);
        if constexpr (WORDS == 3)
            lds[(2 * LDS_DEPTH + slot) * 256] = __float_as_uint(loc);
    }
    DEV void pop(int pos, uint32_t &ref, float &d, float &loc) {
        const uint32_t slot = slot_of((uint32_t)pos);
        ref = lds[(0 * LDS_DEPTH + slot) * 256];
        d = __uint_as_float(lds[(1 * LDS_DEPTH + slot) * 256]);
        if constexpr (WORDS == 3)
            loc = __uint_as_float(lds[(2 * LDS_DEPTH + slot) * 256]);
        if (pos < base) { // the ring is empty: take the frame back from the workspace
            const Rec v = ov[(size_t)pos * stride];
            ref = v.x;
            d = __uint_as_float(v.y);
            if constexpr (WORDS == 3)
                loc = __uint_as_float(v.z);
            base = pos;
        }
    }
    // pop() inside straight-line wave code: every lane reads some valid LDS slot (pos may be negative or stale on lanes
    // that do not `want` the frame), only wanting lanes look at the workspace
    DEV void pop_masked(int pos, bool want, uint32_t &ref, float &d, float &loc) {
        const uint32_t p = (uint32_t)pos < (uint32_t)RT_MAX_STACK ? (uint32_t)pos : 0u;
        const uint32_t slot = slot_of(p);
        ref = lds[(0 * LDS_DEPTH + slot) * 256];
        d = __uint_as_float(lds[(1 * LDS_DEPTH + slot) * 256]);
        if constexpr (WORDS == 3)
            loc = __uint_as_float(lds[(2 * LDS_DEPTH + slot) * 256]);
        // keep the LDS reads where they are: sunk into the branch below they would merge with the global loads into
        // generic flat_loads of a selected pointer
        if constexpr (WORDS == 3)
            asm volatile("" : "+v"(ref), "+v"(d), "+v"(loc));
        else
            asm volatile("" : "+v"(ref), "+v"(d));
        if (want && pos < base) {
            const Rec v = ov[(size_t)pos * stride];
            ref = v.x;
            d = __uint_as_float(v.y);
            if constexpr (WORDS == 3)
                loc = __uint_as_float(v.z);
            base = pos;
        }
    }
};
#define RT_DECLARE_RING_STACK_W(NAME, DEPTH, WORDS, SHARED_ARRAY, OVERFLOW, STRIDE)                                                    \
    RingStackT<(DEPTH), (WORDS)> NAME;                                                                                            \
    NAME.lds = (SHARED_ARRAY) + threadIdx.x;                                                                                      \
    NAME.ov = reinterpret_cast<RingStackT<(DEPTH), (WORDS)>::Rec *>(OVERFLOW) + ((size_t)blockIdx.x * blockDim.x + threadIdx.x); \
    NAME.stride = (STRIDE);                                                                                                       \
    NAME.base = 0
#define RT_DECLARE_RING_STACK(NAME, DEPTH, SHARED_ARRAY, OVERFLOW, STRIDE) RT_DECLARE_RING_STACK_W(NAME, DEPTH, 3, SHARED_ARRAY, OVERFLOW, STRIDE)

#define STACK_LDS_DWORDS (3 * LDS_DEPTH * 256)
#define STACK_LDS_DWORDS_FOR(depth) (3 * (depth) * 256)
#define STACK_LDS_DWORDS_W(depth, words) ((words) * (depth) * 256)
#define RT_DECLARE_STACK(NAME, DEPTH, SHARED_ARRAY)          \
    uint32_t NAME##_ov_ref[RT_MAX_STACK - (DEPTH)];         \
    float NAME##_ov_d[RT_MAX_STACK - (DEPTH)];              \
    float NAME##_ov_loc[RT_MAX_STACK - (DEPTH)];            \
    StackMemT<(DEPTH)> NAME;                                \
    NAME.lds = (SHARED_ARRAY) + threadIdx.x;                \
    NAME.ov_ref = NAME##_ov_ref;                            \
    NAME.ov_d = NAME##_ov_d;                                \
    NAME.ov_loc = NAME##_ov_loc

// ---------------------------------------------------------------------------------------------- closest hit
// BVH::intersect_ray (bvh.h:170-180, 195-235) as a resumable per-lane state machine: one call of trav_step visits
// ONE record — an inner node (both child boxes, 64 B) or one leaf triangle (48 B) — so every lane of the wavefront
// issues exactly one gather per step whatever it is doing, and a lane can be parked between steps while others shade.
//   frame = {far child ref, far entry distance d_far, local best of the ENCLOSING subtree at push time}
//   t_loc = local best t of the subtree being traversed (NaN = no hit yet; fminf ignores NaN operands).
// On pop the far sibling is visited iff the near subtree found nothing or found t > d_far (bvh.h:221): the
// reference prunes against the near subtree's local best only. The global best uses the strict "replace iff existing
// t > new t" rule (bvh.h:132) in DFS order, which equals the nested update_intersection calls.
constexpr uint32_t T_DONE = 0xFFFFFFFEu, T_POP = 0xFFFFFFFDu;
struct Trav {
    V3 o, d, r; // r = 1/d (IEEE) for div_exact_fast
    uint32_t cur;
    int sp;
    float t_loc;
    Hit best;
    bool fast; // div_exact_fast is valid for this ray (see its comment)
    // newest frame (stack position sp-1) cached in registers: a pop followed by a node visit never waits for LDS
    uint32_t top_ref;
    float top_d, top_loc;
};
// The per-ray half of div_exact_fast's preconditions (the per-scene half is DevBvh::fast_ok, checked by the host)
DEV bool ray_fast_ok(const DevBvh &bvh, V3 o, V3 d) {
    const float lo = fminf(fminf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    const float hi = fmaxf(fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    return (bvh.fast_ok != 0u) & (d.x == d.x) & (d.y == d.y) & (d.z == d.z) & (lo >= RANGE_LO) & (hi <= RANGE_HI) & coord_in_fast_range(o.x) &
           coord_in_fast_range(o.y) & coord_in_fast_range(o.z);
}
// the per-ray half alone (stored with a queued ray, WfPath::fast) and the start of a traversal from stored values
DEV bool ray_fast_ok_ray(V3 o, V3 d) {
    const float lo = fminf(fminf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    const float hi = fmaxf(fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    return (d.x == d.x) & (d.y == d.y) & (d.z == d.z) & (lo >= RANGE_LO) & (hi <= RANGE_HI) & coord_in_fast_range(o.x) & coord_in_fast_range(o.y) &
           coord_in_fast_range(o.z);
}
// GB (production traversal, RT_FLAG_GLOBAL_BEST): T.t_loc is the GLOBAL best t so far (+inf before the first hit) and every
// box is culled against it: a child is visited iff !(best.t <= its entry distance). The reference prunes a far child only
// against the near subtree's local best (bvh.h:216-223), so the global rule visits a SUBSET of the reference's nodes and
// returns the same hit unless a triangle's t rounds below its own box's entry distance — measured per scene by
// tests/test_gpu_production.py. Frames shrink to {ref, d_far}: no saved local best.
template <bool GB = false> DEV void trav_init_stored(Trav &T, const DevBvh &bvh, V3 o, V3 d, V3 r, bool ray_ok) {
    T.o = o;
    T.d = d;
    T.r = r;
    T.fast = ray_ok & (bvh.fast_ok != 0u);
    T.cur = (bvh.root == RT_NONE || bvh.n_tris == 0) ? T_DONE : bvh.root;
    T.sp = 0;
    T.t_loc = GB ? RT_INF : RT_NAN;
    if constexpr (GB)
        T.top_loc = 0.0f; // unused by the global-best frames
    T.best = Hit{RT_NONE, 0.f, 0.f, 0.f};
}
DEV void trav_init(Trav &T, const DevBvh &bvh, V3 o, V3 d) {
    T.o = o;
    T.d = d;
    T.r = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    T.fast = ray_fast_ok(bvh, o, d);
    T.cur = (bvh.root == RT_NONE || bvh.n_tris == 0) ? T_DONE : bvh.root;
    T.sp = 0;
    T.t_loc = RT_NAN;
    T.best = Hit{RT_NONE, 0.f, 0.f, 0.f};
}

// Unwind deferred siblings after a leaf or a double miss: the far child of the newest frame is visited iff the near
// subtree found nothing or found t > d_far (bvh.h:221); either way the near result is merged into the enclosing
// subtree's local best. Ends in T_DONE when the stack is empty.
template <class STK> DEV void trav_pop(Trav &T, STK &stk) {
    while (T.cur == T_POP) {
        DIAG(7, 1);
        DIAG_LANES(8);
        if (T.sp == 0) {
            T.cur = T_DONE;
            break;
        }
        --T.sp;
        const uint32_t ref = T.top_ref;
        const float dfar = T.top_d, saved = T.top_loc;
        if (T.sp > 0)
            stk.pop(T.sp - 1, T.top_ref, T.top_d, T.top_loc); // refill the register copy; consumed at the next pop
        const float t_near = T.t_loc;
        T.t_loc = fminf(saved, t_near);
        if (!(t_near <= dfar)) // !has || t_near > d_far (bvh.h:221)
            T.cur = ref;
    }
}

// One record: an inner node (two child boxes) or one triangle of a big leaf. Leaves T.cur == T_POP when the lane has to
// unwind; the caller chooses how (trav_pop: per-lane loop; trav_pop_wave: all lanes of the wave together).
template <bool STATS, bool GB = false, class STK> DEV void trav_step_core(Trav &T, const DevBvh &bvh, STK &stk, float min_dst, LaneStats<STATS> &st) {
    const bool leaf = (T.cur & RT_LEAF_FLAG) != 0;
    const float4 *p = leaf ? reinterpret_cast<const float4 *>(bvh.tris + (T.cur & RT_LEAF_BEGIN_MASK)) : reinterpret_cast<const float4 *>(bvh.nodes + T.cur);
    const float4 r0 = p[0], r1 = p[1], r2 = p[2];
    if (!leaf) {
        const float4 r3 = p[3];
        st.node();
        st.box(2);
        // DevNode: lmin.xyz lmax.xyz rmin.xyz rmax.xyz left right
        const V3 lmin = mk(r0.x, r0.y, r0.z), lmax = mk(r0.w, r1.x, r1.y), rmn = mk(r1.z, r1.w, r2.x), rmx = mk(r2.y, r2.z, r2.w);
        const uint32_t left = __float_as_uint(r3.x), right = __float_as_uint(r3.y);
        float dl, dr;
        bool hl, hr;
        DIAG(2, 1);
        DIAG_LANES(3);
        if (T.fast) {
            hl = box_hit_fast(lmin, lmax, T.o, T.d, T.r, min_dst, dl);
            hr = box_hit_fast(rmn, rmx, T.o, T.d, T.r, min_dst, dr);
        } else { // rare ray: a direction/origin component is 0-adjacent, huge or NaN -> reference arithmetic
            DIAG(6, 1);
            hl = box_hit_exact(lmin, lmax, T.o, T.d, min_dst, dl);
            hr = box_hit_exact(rmn, rmx, T.o, T.d, min_dst, dr);
        }
        if constexpr (GB) { // cull against the global best
            hl = hl && dl < T.t_loc;
            hr = hr && dr < T.t_loc;
        }
        if (hl & hr) {
            uint32_t near = left, far = right;
            float dfar = dr;
            if (dl > dr) { // bvh.h:216 (ties keep left first)
                near = right;
                far = left;
                dfar = dl;
            }
            if (T.sp > 0)
                stk.push(T.sp - 1, T.top_ref, T.top_d, T.top_loc); // spill the previous top
            T.top_ref = far;
            T.top_d = dfar;
            if constexpr (!GB) {
                T.top_loc = T.t_loc;
                T.t_loc = RT_NAN;
            }
            ++T.sp;
            T.cur = near;
        } else if (hl) {
            T.cur = left;
        } else if (hr) {
            T.cur = right;
        } else {
            T.cur = T_POP;
        }
    } else {
        DIAG(4, 1);
        DIAG_LANES(5);
        // DevTri: a.xyz v.xyz u.xyz prim flags pad
        const uint32_t flags = __float_as_uint(r2.z);
        if (flags & 2u)
            st.node(); // first triangle of its leaf: one BVH::intersect_ray invocation on the leaf node
        st.tri();
        V3 xs;
        if (tri_hit(mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), T.o, T.d, min_dst, xs)) {
            if (T.best.k == RT_NONE || T.best.t > xs.z) {
                T.best.k = T.cur & RT_LEAF_BEGIN_MASK;
                T.best.b = xs.x;
                T.best.c = xs.y;
                T.best.t = xs.z;
            }
            T.t_loc = fminf(T.t_loc, xs.z);
        }
        T.cur = (flags & 1u) ? T_POP : T.cur + 1;
    }
}
template <bool STATS, class STK> DEV void trav_step(Trav &T, const DevBvh &bvh, STK &stk, float min_dst, LaneStats<STATS> &st) {
    trav_step_core<STATS, false>(T, bvh, stk, min_dst, st);
    trav_pop(T, stk);
}

// The same two operations for the wavefront kernel's hot loop, written so that a wave runs one straight-line sequence
// (selects instead of per-lane branches: no exec-mask nesting and no register copies at control-flow joins).
//   trav_step_inner_fast : trav_step_core for lanes the caller knows to be on an inner node with T.fast
//   trav_pop_wave        : trav_pop for every lane in T_POP at once; lanes leave the loop as a wave
template <bool STATS, bool GB = false, class STK> DEV void trav_step_inner_fast(Trav &T, const DevBvh &bvh, STK &stk, float min_dst, LaneStats<STATS> &st) {
    const float4 *p = reinterpret_cast<const float4 *>(bvh.nodes + T.cur);
    const float4 r0 = p[0], r1 = p[1], r2 = p[2];
    const float2 r3 = *reinterpret_cast<const float2 *>(p + 3);
    st.node();
    st.box(2);
    const uint32_t left = __float_as_uint(r3.x), right = __float_as_uint(r3.y);
    float dl, dr;
    bool hl = box_hit_fast(mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), T.o, T.d, T.r, min_dst, dl);
    bool hr = box_hit_fast(mk(r1.z, r1.w, r2.x), mk(r2.y, r2.z, r2.w), T.o, T.d, T.r, min_dst, dr);
    if constexpr (GB) { // cull against the global best
        hl &= dl < T.t_loc;
        hr &= dr < T.t_loc;
    }
    const bool both = hl & hr;
    const bool swap = dl > dr; // bvh.h:216 (ties keep left first)
    if (both & (T.sp > 0))
        stk.push(T.sp - 1, T.top_ref, T.top_d, T.top_loc); // spill the previous top
    T.top_ref = both ? (swap ? left : right) : T.top_ref;
    T.top_d = both ? (swap ? dl : dr) : T.top_d;
    if constexpr (!GB) {
        T.top_loc = both ? T.t_loc : T.top_loc;
        T.t_loc = both ? RT_NAN : T.t_loc;
    }
    T.sp += both ? 1 : 0;
    T.cur = both ? (swap ? right : left) : (hl ? left : (hr ? right : T_POP));
}
// The same node step on a record the caller already holds (wf_extend_packet: one scalar fetch serves every lane standing on
// the node). Guarded rays (T.fast false) take the reference division; the bookkeeping is trav_step_inner_fast's.
template <bool STATS, bool GB = false, class STK>
DEV void trav_inner_apply(Trav &T, STK &stk, V3 lmin, V3 lmax, V3 rmn, V3 rmx, uint32_t left, uint32_t right, float min_dst, LaneStats<STATS> &st) {
    st.node();
    st.box(2);
    float dl, dr;
    bool hl, hr;
    if (T.fast) {
        hl = box_hit_fast(lmin, lmax, T.o, T.d, T.r, min_dst, dl);
        hr = box_hit_fast(rmn, rmx, T.o, T.d, T.r, min_dst, dr);
    } else {
        hl = box_hit_exact(lmin, lmax, T.o, T.d, min_dst, dl);
        hr = box_hit_exact(rmn, rmx, T.o, T.d, min_dst, dr);
    }
    if constexpr (GB) { // cull against the global best
        hl = hl && dl < T.t_loc;
        hr = hr && dr < T.t_loc;
    }
    const bool both = hl & hr;
    const bool swap = dl > dr; // bvh.h:216 (ties keep left first)
    if (both & (T.sp > 0))
        stk.push(T.sp - 1, T.top_ref, T.top_d, T.top_loc); // spill the previous top
    T.top_ref = both ? (swap ? left : right) : T.top_ref;
    T.top_d = both ? (swap ? dl : dr) : T.top_d;
    if constexpr (!GB) {
        T.top_loc = both ? T.t_loc : T.top_loc;
        T.t_loc = both ? RT_NAN : T.t_loc;
    }
    T.sp += both ? 1 : 0;
    T.cur = both ? (swap ? right : left) : (hl ? left : (hr ? right : T_POP));
}
// Scalar (s_load) reads of scene constants at a wave-uniform address: the constant address space tells the compiler that
// nothing in the kernel writes them.
typedef float F4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) F4v *ConstF4;
DEV ConstF4 as_const_f4(const void *p) { return (ConstF4)(unsigned long long)p; }
// minimum of a value over the 64 lanes of a fully active wave (row shifts, then the two row broadcasts of gfx9 DPP)
DEV uint32_t wave_min_u32(uint32_t v) {
    const int id = -1; // 0xFFFFFFFF: what a lane without a source reads
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x111, 0xF, 0xF, false)); // row_shr:1
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x112, 0xF, 0xF, false)); // row_shr:2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x114, 0xF, 0xF, false)); // row_shr:4
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x118, 0xF, 0xF, false)); // row_shr:8 -> lane 15 of a row = row minimum
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x142, 0xA, 0xF, false)); // row_bcast:15 into rows 1 and 3
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x143, 0xC, 0xF, false)); // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// one unwind step for every lane in T_POP, as straight-line wave code (lanes in other states pass through unchanged)
template <bool GB = false, class STK> DEV void trav_pop_once(Trav &T, STK &stk) {
    DIAG(7, 1);
    DIAG(27, (unsigned long long)__popcll(__ballot(T.cur == T_POP)));
    const bool pop = T.cur == T_POP;
    const bool go = pop & (T.sp != 0);
    const int nsp = T.sp - 1;
    const bool refill = go & (nsp > 0);
    uint32_t n_ref;
    float n_d, n_loc = 0.0f;
    stk.pop_masked(nsp - 1, refill, n_ref, n_d, n_loc);
    const float t_near = T.t_loc;
    // reference: !has || t_near > d_far against the NEAR subtree's local best (bvh.h:221); GB: against the global best
    const bool visit = !(t_near <= T.top_d);
    T.cur = pop ? (go ? (visit ? T.top_ref : T_POP) : T_DONE) : T.cur;
    if constexpr (!GB)
        T.t_loc = go ? fminf(T.top_loc, t_near) : t_near;
    T.sp = go ? nsp : T.sp;
    T.top_ref = refill ? n_ref : T.top_ref;
    T.top_d = refill ? n_d : T.top_d;
    if constexpr (!GB)
        T.top_loc = refill ? n_loc : T.top_loc;
}
template <bool GB = false, class STK> DEV void trav_pop_wave(Trav &T, STK &stk) {
    while (__ballot(T.cur == T_POP) != 0ull)
        trav_pop_once<GB>(T, stk);
}
// bvh_mix_dist::pdf (raytracer.h:363-375) = BVH::foreach_intersection (bvh.h:237-260) over the light BVH summing
// triangle_dist::pdf_at (raytracer.h:255-261) in DFS order (node objects, left subtree, right subtree).
// Where the light BVH is read from: its device arrays, or the copy wf_shade stages in LDS when the tree is small
// (DevBvh::lds_inner). The records are the same 16-byte pieces either way.
struct LightTabs {
    const float4 *nodes, *tris, *aux; // DevNode = 4, DevTri = 3, DevLightAux = 1 pieces per record
};
DEV LightTabs light_tabs_global(const DevScene &S) {
    return LightTabs{reinterpret_cast<const float4 *>(S.lights.nodes), reinterpret_cast<const float4 *>(S.lights.tris), reinterpret_cast<const float4 *>(S.light_aux)};
}
template <bool STATS, class STK> DEV float lights_pdf(const DevScene &S, const LightTabs &LT, V3 x, V3 d, STK &stk, LaneStats<STATS> &st) {
    const DevBvh &bvh = S.lights;
    st.lq();
    float res = 0;
    if (bvh.root != RT_NONE && bvh.n_tris != 0) {
        uint32_t cur = bvh.root;
        int sp = 0;
        const bool fast = ray_fast_ok(bvh, x, d); // same exact-quotient shortcut as the closest-hit traversal
        const V3 r = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        while (cur != T_DONE) {
            const bool leaf = (cur & RT_LEAF_FLAG) != 0;
            const float4 *p = leaf ? LT.tris + 3u * (cur & RT_LEAF_BEGIN_MASK) : LT.nodes + 4u * cur;
            const float4 r0 = p[0], r1 = p[1], r2 = p[2];
            if (!leaf) {
                const float4 r3 = p[3];
                st.lnode();
                st.lbox(2);
                const uint32_t left = __float_as_uint(r3.x), right = __float_as_uint(r3.y);
                float dl, dr;
                bool hl, hr;
                if (fast) {
                    hl = box_hit_fast(mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), x, d, r, EPS, dl);
                    hr = box_hit_fast(mk(r1.z, r1.w, r2.x), mk(r2.y, r2.z, r2.w), x, d, r, EPS, dr);
                } else {
                    hl = box_hit_exact(mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), x, d, EPS, dl);
                    hr = box_hit_exact(mk(r1.z, r1.w, r2.x), mk(r2.y, r2.z, r2.w), x, d, EPS, dr);
                }
                if (hl & hr) {
                    stk.push_ref(sp++, right);
                    cur = left;
                } else if (hl) {
                    cur = left;
                } else if (hr) {
                    cur = right;
                } else {
                    cur = T_POP;
                }
            } else {
                const uint32_t k = cur & RT_LEAF_BEGIN_MASK;
                const uint32_t flags = __float_as_uint(r2.z);
                if (flags & 2u)
                    st.lnode();
                st.ltri();
                V3 xs;
                if (tri_hit(mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), x, d, EPS, xs)) {
                    st.lhit();
                    const float4 aux = LT.aux[k];
                    V3 y = x + d * xs.z;  // ray.at(t)
                    V3 dir = norm(y - x); // raytracer.h:259
                    float mult = len2(x - y) / __builtin_fabsf(dot(dir, mk(aux.x, aux.y, aux.z))); // :79-84
                    res += mult / aux.w;
                }
                cur = (flags & 1u) ? T_POP : cur + 1;
            }
            if (cur == T_POP)
                cur = sp ? stk.pop_ref(--sp) : T_DONE;
            SD_STAMP(SD_LPDF); // (development census: one stamp per loop trip, with the lanes still walking the light BVH)
        }
    }
    return res / (float)bvh.n_tris; // res / bvh->objects.size()
}


// ---------------------------------------------------------------------------------------------- textures
// wrap_repeat geometry.h:517-519: std::fmod(std::fmod(x, 1) + 1, 1) evaluated in DOUBLE (float, int -> double
// overload); fmod(x, 1) == x - trunc(x) exactly.
DEV float wrap_repeat(float x) {
    double xd = (double)x;
    double f = xd - __builtin_trunc(xd);
    double g = f + 1.0;
    double h = g - __builtin_trunc(g);
    return (float)h;
}
DEV int mod_inc(int x, int mod) { return x == mod - 1 ? 0 : x + 1; }

enum { TEX_DEFAULT_WHITE = 0, TEX_DEFAULT_NORMAL_UP = 1 };

// Texture::sample (geometry.h:545-575). Texels are RGBA8; k/255.0f and powf(k/255.0f, 2.2f) come from the two
// 256-entry tables staged in LDS (bit-identical to the per-lookup arithmetic of geometry.h:525-527, 593-594).
template <bool STATS>
DEV C4 tex_sample(const DevScene &S, int32_t tex, int dflt, float u, float v, bool gamma, const float *s_lin, const float *s_gam, LaneStats<STATS> &st) {
    if (tex < 0) {
        if (dflt == TEX_DEFAULT_WHITE)
            return C4{1, 1, 1, 1}; // WHITE_TEXTURE geometry.h:601
        return C4{0.5f, 0.5f, 1, 0}; // NORMAL_UP geometry.h:602
    }
    DevTexture T;
    {
        const uint4 *tp = reinterpret_cast<const uint4 *>(S.textures + tex);
        const uint4 t0 = tp[0], t1 = tp[1];
        T.width = t0.x, T.height = t0.y, T.offset = t0.z, T.count = t0.w;
        T.stride = t1.x, T.tiles_x = t1.y, T.tw_log = t1.z, T.th_log = t1.w;
    }
    if (T.count == 1) { // 1x1 fast path returns the texel WITHOUT gamma (geometry.h:548-550)
        uint32_t p = S.texels[T.offset];
        return C4{s_lin[p & 255u], s_lin[(p >> 8) & 255u], s_lin[(p >> 16) & 255u], s_lin[p >> 24]};
    }
    float tx = wrap_repeat(u) * (float)T.width;
    float ty = wrap_repeat(v) * (float)T.height;
    int px = (int)tx;
    int py = (int)ty;
    float dx = tx - (float)px;
    float dy = ty - (float)py;
    const int w = (int)T.width, h = (int)T.height;
    int x0 = px, x1 = mod_inc(px, w), y0 = py, y1 = mod_inc(py, h);
    // tiled address of texel (x, y) of this view
    auto at = [&](int x, int y) {
        const uint32_t tile = ((uint32_t)y >> T.th_log) * T.tiles_x + ((uint32_t)x >> T.tw_log);
        const uint32_t within = (((uint32_t)y & ((1u << T.th_log) - 1u)) << T.tw_log) | ((uint32_t)x & ((1u << T.tw_log) - 1u));
        return S.texels[T.offset + ((tile << (T.tw_log + T.th_log)) + within) * T.stride];
    };
    uint32_t q00, q01, q10, q11;
    if (px < w && py < h) {
        q00 = at(x0, y0), q01 = at(x0, y1), q10 = at(x1, y0), q11 = at(x1, y1);
    } else {
        // wrap_repeat rounded up to 1.0f: the reference indexes row-major position x + y*w past the row / the image
        // (geometry.h:556-563); as in the oracle the flat index is clamped for memory safety only, then located
        // Located WITHOUT an integer division (its expansion was the kernel's register-pressure peak, on a path almost no
        // lookup takes): here 0 <= x <= w + 1 and 0 <= y <= h + 1, so the flat index i = x + y * w lies in row y + x / w with
        // x / w in {0, 1, 2}, and an index beyond the last texel is the last texel (w - 1, h - 1).
        auto flat = [&](int x, int y) {
            const int over = x >= 2 * w ? 2 : (x >= w ? 1 : 0);
            int row = y + over, col = x - over * w;
            if (row >= h) { // i > last
                row = h - 1;
                col = w - 1;
            }
            return at(col, row);
        };
        q00 = flat(x0, y0), q01 = flat(x0, y1), q10 = flat(x1, y0), q11 = flat(x1, y1);
    }
    st.texels(4);
    const float *rgb = gamma ? s_gam : s_lin;
    auto dec = [&](uint32_t p) { return C4{rgb[p & 255u], rgb[(p >> 8) & 255u], rgb[(p >> 16) & 255u], s_lin[p >> 24]}; };
    C4 p00 = dec(q00), p01 = dec(q01), p10 = dec(q10), p11 = dec(q11);
    return (1 - dx) * ((1 - dy) * p00 + dy * p01) + dx * ((1 - dy) * p10 + dy * p11);
}

// The same four lookups when all of a material's textures are members of ONE interleaved view set (DevMaterial::tex_set): equal
// size and tiling, record = {colour, emissive, metallic-roughness, normal} texel at one position. Texture::sample's address
// arithmetic (wrap, truncation, neighbours: geometry.h:545-563) is the same for every member, so it is done once, and each of the
// four neighbouring records is ONE 16-byte load instead of a 4-byte load per member; every member is then decoded and blended
// with exactly tex_sample's operations in tex_sample's order, so each result is bit-identical to the slot-by-slot path.
struct TexSet {
    C4 color, emissive, mr, normal;
};
template <bool STATS>
DEV TexSet tex_sample_set(const DevScene &S, int32_t view, uint32_t info, float u, float v, const float *s_lin, const float *s_gam, LaneStats<STATS> &st) {
    DevTexture T;
    {
        const uint4 *tp = reinterpret_cast<const uint4 *>(S.textures + view);
        const uint4 t0 = tp[0], t1 = tp[1];
        T.width = t0.x, T.height = t0.y, T.offset = t0.z, T.count = t0.w;
        T.stride = t1.x, T.tiles_x = t1.y, T.tw_log = t1.z, T.th_log = t1.w;
    }
    const uint32_t base = T.offset - ((info >> 4) & 3u); // dword 0 of record (0, 0)
    float tx = wrap_repeat(u) * (float)T.width;
    float ty = wrap_repeat(v) * (float)T.height;
    int px = (int)tx;
    int py = (int)ty;
    float dx = tx - (float)px;
    float dy = ty - (float)py;
    const int w = (int)T.width, h = (int)T.height;
    int x0 = px, x1 = mod_inc(px, w), y0 = py, y1 = mod_inc(py, h);
    auto at = [&](int x, int y) {
        const uint32_t tile = ((uint32_t)y >> T.th_log) * T.tiles_x + ((uint32_t)x >> T.tw_log);
        const uint32_t within = (((uint32_t)y & ((1u << T.th_log) - 1u)) << T.tw_log) | ((uint32_t)x & ((1u << T.tw_log) - 1u));
        return *reinterpret_cast<const uint4 *>(S.texels + base + ((tile << (T.tw_log + T.th_log)) + within) * 4u);
    };
    uint4 q00, q01, q10, q11;
    if (px < w && py < h) {
        q00 = at(x0, y0), q01 = at(x0, y1), q10 = at(x1, y0), q11 = at(x1, y1);
    } else { // wrap_repeat rounded up to 1.0f: see tex_sample
        auto flat = [&](int x, int y) {
            const int over = x >= 2 * w ? 2 : (x >= w ? 1 : 0);
            int row = y + over, col = x - over * w;
            if (row >= h) {
                row = h - 1;
                col = w - 1;
            }
            return at(col, row);
        };
        q00 = flat(x0, y0), q01 = flat(x0, y1), q10 = flat(x1, y0), q11 = flat(x1, y1);
    }
    auto blend = [&](uint32_t a00, uint32_t a01, uint32_t a10, uint32_t a11, bool gamma) {
        st.texels(4);
        const float *rgb = gamma ? s_gam : s_lin;
        auto dec = [&](uint32_t p) { return C4{rgb[p & 255u], rgb[(p >> 8) & 255u], rgb[(p >> 16) & 255u], s_lin[p >> 24]}; };
        C4 p00 = dec(a00), p01 = dec(a01), p10 = dec(a10), p11 = dec(a11);
        return (1 - dx) * ((1 - dy) * p00 + dy * p01) + dx * ((1 - dy) * p10 + dy * p11);
    };
    TexSet r;
    r.normal = (info & 8u) ? blend(q00.w, q01.w, q10.w, q11.w, false) : C4{0.5f, 0.5f, 1, 0}; // NORMAL_UP geometry.h:602
    r.mr = (info & 4u) ? blend(q00.z, q01.z, q10.z, q11.z, false) : C4{1, 1, 1, 1};             // WHITE_TEXTURE geometry.h:601
    r.color = (info & 1u) ? blend(q00.x, q01.x, q10.x, q11.x, true) : C4{1, 1, 1, 1};
    r.emissive = (info & 2u) ? blend(q00.y, q01.y, q10.y, q11.y, true) : C4{1, 1, 1, 1};
    return r;
}

// ---------------------------------------------------------------------------------------------- shading record
struct Surf { // ray_intersection_info bvh.h:18-29
    V3 normal, shading_normal;
    C4 color;
    V3 emission;
    float metallic, roughness, ior;
};

// to_intersection_info bvh.h:80-121
template <bool STATS>
DEV Surf make_surf(const DevScene &S, const Hit &h, V3 ro, V3 rd, const float *s_lin, const float *s_gam, LaneStats<STATS> &st) {
    if (h.k & RT_PRIM_FLAG) { // analytic primitive: geometric normal only, untextured material (rt_primspec.h)
        const rt_primitive_desc p = load_prim(S.prims, h.k & ~RT_PRIM_FLAG);
        const float oo[3] = {ro.x, ro.y, ro.z}, dd[3] = {rd.x, rd.y, rd.z};
        float t, n[3] = {0.f, 0.f, 1.f};
        (void)rt_prim_intersect(&p, oo, dd, 1e-4f, &t, n); // same inputs as the cast -> same root, same normal
        DevMaterial m;
        {
            const float4 *mp = reinterpret_cast<const float4 *>(S.materials + p.material_id);
            float4 *q = reinterpret_cast<float4 *>(&m);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                q[i] = mp[i];
        }
        st.shaded();
        Surf s;
        s.normal = s.shading_normal = mk(n[0], n[1], n[2]);
        s.color = C4{m.color[0], m.color[1], m.color[2], m.color[3]};
        s.emission = mk(m.emission[0], m.emission[1], m.emission[2]);
        s.metallic = m.metallic;
        s.roughness = m.roughness;
        s.ior = m.ior;
        return s;
    }
    DevAttr at;
    {
        const float4 *p = reinterpret_cast<const float4 *>(S.attrs + h.k);
        float4 *q = reinterpret_cast<float4 *>(&at);
#pragma unroll
        for (int i = 0; i < 7; ++i)
            q[i] = p[i];
    }
    // The material record is read in two halves: the texture slots and scalar factors now, colour / emission / roughness only
    // AFTER the four texture lookups (their registers would otherwise sit idle through the kernel's register-pressure peak).
    // The second read hits the line the first one brought in.
    DevMaterial m;
    const float4 *mat_p = reinterpret_cast<const float4 *>(S.materials + at.material);
    {
        float4 *q = reinterpret_cast<float4 *>(&m);
        q[2] = mat_p[2];
        q[3] = mat_p[3];
    }
    SD_STAMP(SD_ATTR);
    const float b = h.b, c = h.c;
    const float w0 = (1 - b - c); // triangle::interop geometry.h:497-502
    V3 normal = ld3(at.gn);
    bool is_inside = dot(normal, rd) > 0;
    V3 smooth = norm(ld3(at.n) * w0 + ld3(at.n + 3) * b + ld3(at.n + 6) * c);
    if (dot(normal, smooth) < 0)
        smooth = -smooth;
    float tu = at.uv[0] * w0 + at.uv[2] * b + at.uv[4] * c;
    float tv = at.uv[1] * w0 + at.uv[3] * b + at.uv[5] * c;
    V3 tangent = norm(ld3(at.tg) * w0 + ld3(at.tg + 3) * b + ld3(at.tg + 6) * c);
    V3 bitangent = crs(smooth, tangent);
    C4 nt, mr, ct, et;
    V3 shading;
    if (m.tex_set >= 0) { // all lookups of this material in one interleaved set: addresses once, one 16-byte load per neighbour
        const TexSet ts = tex_sample_set(S, m.tex_set, m.tex_set_info, tu, tv, s_lin, s_gam, st);
        nt = ts.normal, mr = ts.mr, ct = ts.color, et = ts.emissive;
        V3 normal_loc = norm(mk(nt.r, nt.g, nt.b) * 2 - 1);
        shading = norm(transform3(normal_loc, tangent, bitangent, smooth));
    } else {
        nt = tex_sample(S, m.normal_tex, TEX_DEFAULT_NORMAL_UP, tu, tv, false, s_lin, s_gam, st); // sample_normal geometry.h:577-582
        V3 normal_loc = norm(mk(nt.r, nt.g, nt.b) * 2 - 1);
        shading = norm(transform3(normal_loc, tangent, bitangent, smooth));
        mr = tex_sample(S, m.mr_tex, TEX_DEFAULT_WHITE, tu, tv, false, s_lin, s_gam, st); // geometry.h:623-626
        ct = tex_sample(S, m.color_tex, TEX_DEFAULT_WHITE, tu, tv, true, s_lin, s_gam, st); // :615-617
        et = tex_sample(S, m.emissive_tex, TEX_DEFAULT_WHITE, tu, tv, true, s_lin, s_gam, st); // :619-621
    }
    st.shaded();
    SD_STAMP(SD_TEX);
    {
        asm volatile("" : "+v"(mat_p)); // not before this point
        float4 *q = reinterpret_cast<float4 *>(&m);
        q[0] = mat_p[0];
        q[1] = mat_p[1];
    }
    Surf s;
    s.normal = is_inside ? -normal : normal;
    s.shading_normal = is_inside ? -shading : shading;
    s.color = C4{m.color[0], m.color[1], m.color[2], m.color[3]} * ct;
    s.emission = mk(m.emission[0], m.emission[1], m.emission[2]) * mk(et.r, et.g, et.b);
    s.metallic = m.metallic * mr.b;
    s.roughness = m.roughness * mr.g;
    s.ior = m.ior;
    return s;
}

// ---------------------------------------------------------------------------------------------- sampling + BRDF
template <class R> DEV V3 sphere_uniform(R &rng) { // raytracer.h:94-105
    float z = uniform_real(rng, -1.0f, 1.0f);
    float co_z = __builtin_sqrtf(rmax(0.0f, 1 - z * z));
    float phi = uniform_real(rng, 0.0f, 2 * PI_F);
    float s, c;
    rt_sincos_libm(phi, &s, &c); // std::cos / std::sin on floats = glibc cosf / sinf, restated bit for bit (rt_devspec.h)
    return {co_z * c, co_z * s, z};
}
DEV V3 halfway(V3 in_dir, V3 out_dir) { return norm(out_dir - in_dir); } // :131-134
DEV V3 choose_local_x(V3 n) { // :208-219
    V3 res{1, 1, 1};
    if (__builtin_fabsf(n.x) > 0.5f)
        res.x -= dot(res, n) / n.x;
    else if (__builtin_fabsf(n.y) > 0.5f)
        res.y -= dot(res, n) / n.y;
    else
        res.z -= dot(res, n) / n.z;
    return norm(res);
}
template <class R> DEV V3 vndf_sample(R &rng, float roughness, V3 in_dir, V3 normal) { // :140-173
    V3 nx = choose_local_x(normal);
    V3 ny = crs(normal, nx);
    V3 v = -norm(mk(dot(nx, in_dir), dot(ny, in_dir), dot(normal, in_dir)));
    V3 vh = norm(mk(roughness, roughness, 1) * v);
    float lensq = vh.x * vh.x + vh.y * vh.y;
    V3 T1 = lensq > 0 ? mk(-vh.y, vh.x, 0) / __builtin_sqrtf(lensq) : mk(1, 0, 0);
    V3 T2 = crs(vh, T1);
    float r = __builtin_sqrtf(uniform_real(rng, 0, 1));
    float phi = 2.0f * PI_F * uniform_real(rng, 0, 1);
    float sn, cs;
    rt_sincos_libm(phi, &sn, &cs);
    float t1 = r * cs;
    float t2 = r * sn;
    float s = 0.5f * (1.0f + vh.z);
    t2 = (1.0f - s) * __builtin_sqrtf(1.0f - pow2(t1)) + s * t2;
    V3 nh = transform3(mk(t1, t2, __builtin_sqrtf(rmax(0.0f, 1.0f - pow2(t1) - pow2(t2)))), T1, T2, vh);
    V3 ne = norm(mk(roughness * nh.x, roughness * nh.y, rmax(0.0f, nh.z)));
    V3 res_n = norm(transform3(ne, nx, ny, normal));
    return in_dir - 2 * res_n * dot(in_dir, res_n); // reflect geometry.h:36-40
}
DEV float vndf_pdf(float roughness, V3 in_dir, V3 normal, V3 dir) { // :175-206
    V3 nx = choose_local_x(normal);
    V3 ny = crs(normal, nx);
    V3 v = -mk(dot(nx, in_dir), dot(ny, in_dir), dot(normal, in_dir));
    V3 nv = halfway(in_dir, dir);
    V3 n = mk(dot(nx, nv), dot(ny, nv), dot(normal, nv));
    float vdn = dot(v, n);
    if (vdn <= 0)
        return 0;
    float vx = v.x * roughness, vy = v.y * roughness;
    float lambda = (-1 + __builtin_sqrtf(1 + (vx * vx + vy * vy) / pow2(v.z))) / 2;
    float g1 = 1 / (1 + lambda);
    float dn = 1 / PI_F / roughness / roughness / pow2(len2(n / mk(roughness, roughness, 1)));
    float dv = g1 * vdn * dn / rmax(EPS, v.z);
    return dv / 4 / vdn;
}
DEV float heaviside(float x) { return x > 0 ? 1.0f : 0.0f; }
DEV float specular_brdf(float alpha, V3 in_dir, V3 out_dir, V3 normal) { // :273-293
    V3 h = halfway(in_dir, out_dir);
    float ndh = dot(normal, h);
    float d = pow2(alpha) * heaviside(ndh) / PI_F / pow2(pow2(ndh) * (pow2(alpha) - 1) + 1);
    float ndo = dot(normal, out_dir);
    float ndi = dot(normal, -in_dir);
    float div1 = (__builtin_fabsf(ndo) + __builtin_sqrtf(pow2(alpha) + (1 - pow2(alpha)) * pow2(ndo)));
    float div2 = (__builtin_fabsf(ndi) + __builtin_sqrtf(pow2(alpha) + (1 - pow2(alpha)) * pow2(ndi)));
    float v = heaviside(dot(h, out_dir)) * heaviside(dot(h, -in_dir)) / div1 / div2;
    return v * d;
}
DEV V3 pbr_brdf(V3 in_dir, V3 out_dir, const Surf &ii) { // :300-343
    V3 res{0, 0, 0};
    V3 base = mk(ii.color.r, ii.color.g, ii.color.b);
    float alpha = pow2(rmax(ii.roughness, MIN_ROUGHNESS));
    float sp = specular_brdf(alpha, in_dir, out_dir, ii.shading_normal);
    V3 spec = mk(sp, sp, sp);
    float VdotH = dot(-in_dir, halfway(in_dir, out_dir));
    float fw = pow5(1 - __builtin_fabsf(VdotH));
    if (ii.metallic < 1) {
        V3 diffuse = base / PI_F;
        float f0 = pow2((1 - ii.ior) / (1 + ii.ior));
        float fr = f0 + (1 - f0) * fw;
        V3 dielectric = diffuse * (1 - fr) + spec * fr;
        res = res + (1 - ii.metallic) * dielectric;
    }
    if (ii.metallic > 0) {
        V3 metal = spec * (base + (1 - base) * fw);
        res = res + ii.metallic * metal;
    }
    return res;
}

// Scene::bg_at (scene.h:83-89): bg_color * bg.sample({x, y}, 2.2f).rgb(). With the default 1x1 WHITE_TEXTURE (USE_ENV_MAP = false,
// config.h:37) Texture::sample returns its only texel before looking at the coordinates, so nothing is evaluated. With an environment
// map the direction goes through the reference's own atan2f / asinf (rt_devspec.h rt_bg_uv) and the ordinary texture lookup with gamma.
// ENV = false compiles the lookup out (wf_shade's default instantiations keep their register budget; the launcher picks ENV by scene).
template <bool STATS, bool ENV = true> DEV V3 bg_at(const DevScene &S, V3 dir, const float *s_lin, const float *s_gam, LaneStats<STATS> &st) {
    if (!ENV || S.bg_tex < 0)
        return ld3(S.bg) * mk(1, 1, 1);
    float u, v;
    rt_bg_uv(dir.x, dir.y, dir.z, &u, &v);
    const C4 c = tex_sample<STATS>(S, S.bg_tex, TEX_DEFAULT_WHITE, u, v, true, s_lin, s_gam, st);
    return ld3(S.bg) * mk(c.r, c.g, c.b);
}

// ---- work tickets of the persistent closest-hit kernels. One head for the whole queue, or (RT_XCD_TICKETS) eight: the queue — in ray order, i.e.
// sorted by origin cell and direction when a sort ran — is cut into eight contiguous parts and a block starts on part blockIdx.x % 8. Blocks b and
// b + 8 are observed to share an XCD (MI355X_MICROARCH.md, workgroup dispatch), so the rays one XCD's L2 serves are neighbours in that order; a
// block whose part has run dry moves on to the next one. Placement changes speed only: every position is handed out exactly once either way.
// Measured (profiles/r03_variants.txt item 16): S-10M production 250.7 -> 256.2 Msamples/s, parity +0.5 %; S-sponza (cache resident) unchanged.
#ifndef RT_XCD_TICKETS
#define RT_XCD_TICKETS 1
#endif
struct TicketState {
    uint32_t part, tried; // wave-uniform
};
DEV TicketState ticket_init() { return TicketState{blockIdx.x & 7u, 0u}; }
// next range [q_lo, q_hi) of at most `chunk` queue positions; false = the whole queue has been handed out
DEV bool ticket_take(uint32_t *counters, uint32_t n_in, uint32_t chunk, TicketState &ts, uint32_t &q_lo, uint32_t &q_hi) {
#if RT_XCD_TICKETS
    const uint32_t n_chunks = (n_in + chunk - 1u) / chunk;
    for (;;) {
        if (ts.tried >= 8u)
            return false;
        const uint32_t c0 = (uint32_t)(((unsigned long long)n_chunks * ts.part) >> 3), c1 = (uint32_t)(((unsigned long long)n_chunks * (ts.part + 1u)) >> 3);
        uint32_t t = 0;
        if ((threadIdx.x & 63u) == 0u)
            t = atomicAdd(counters + WF_CNT_XCD + ts.part * WF_CNT_XCD_STRIDE, 1u);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (c0 + t < c1) {
            q_lo = (c0 + t) * chunk;
            q_hi = q_lo + chunk < n_in ? q_lo + chunk : n_in;
            return true;
        }
        ts.part = (ts.part + 1u) & 7u; // this part is used up: help the next one
        ++ts.tried;
    }
#else
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0u)
        base = atomicAdd(counters + WF_CNT_TICKET, chunk);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    q_lo = base < n_in ? base : n_in;
    q_hi = base + chunk < n_in ? base + chunk : n_in;
    return q_lo != q_hi;
#endif
}

// ---------------------------------------------------------------------------------------------- one shade() level
// trace_ray's hit / miss branch (raytracer.h:602-604) + shade (raytracer.h:555-591) for ONE cast result, without the
// recursion: the caller owns depth bookkeeping and the (emission, scale) fold stack.
//   terminal : the path ends here and contributes `term` to the innermost pending frame
//   push     : a scattering event happened: push (emission, scl) and continue with the ray (nro, nrd)
//   neither  : stochastic alpha pass-through (:559-561): continue with (nro, nrd), no frame
// RNG draw order is the reference's: alpha coin, technique coin, then the sampler's own draws.
// Which sampler the NEXT shade() of a path will run, from the generator state the path record stores (taken by value): the alpha coin
// is drawn first whatever the material (:559), the technique coin second (:565), mix_dist's pick third (:386). 0 VNDF, 1 cosine, 2 light
// triangle. A scheduling hint for wf_shade's lane assignment only: a miss or an alpha pass-through never gets that far, and nothing
// computed depends on it.
template <class R> DEV uint32_t next_shade_class(R rng, bool has_lights) {
    (void)uniform_real(rng, 0.0f, 1.0f);
    if (uniform_real(rng, 0.0f, 1.0f) <= VNDF_FACTOR)
        return 0u;
    return (!has_lights || rng.below(2) == 0u) ? 1u : 2u;
}
struct ShadeResult {
    bool terminal, push;
    V3 term, emission, scl, nro, nrd;
};
template <class R, bool STATS, bool ENV = true, class STK>
DEV ShadeResult shade_hit(const DevScene &S, const LightTabs &LT, const Hit &h, V3 ro, V3 rd, R &rng, bool has_lights, STK &stk, const float *s_lin,
                          const float *s_gam, LaneStats<STATS> &st) {
    ShadeResult out;
    out.terminal = false;
    out.push = false;
    out.term = out.emission = out.scl = out.nro = mk(0, 0, 0);
    out.nrd = mk(0, 0, 1);
    if (h.k == RT_NONE) {
        out.terminal = true;
        out.term = bg_at<STATS, ENV>(S, rd, s_lin, s_gam, st);
        return out;
    }
    const Surf ii = make_surf<STATS>(S, h, ro, rd, s_lin, s_gam, st);
    const V3 pos = ro + rd * h.t;                          // ray.at(t)
    if (!(uniform_real(rng, 0.0f, 1.0f) <= ii.color.a)) { // !coin(alpha) :559-561
        out.nro = pos;
        out.nrd = rd;
        return out;
    }
    SD_STAMP(SD_ALPHA);
    const float vr = pow2(rmax(ii.roughness, MIN_ROUGHNESS)); // :563-564
    V3 dir;
    if (uniform_real(rng, 0.0f, 1.0f) <= VNDF_FACTOR) { // :565
        dir = vndf_sample(rng, vr, rd, ii.shading_normal);
        SD_STAMP(SD_S_VNDF);
    } else if (!has_lights) { // dir_dist = cosine_dist (:449)
        dir = norm(ii.normal + sphere_uniform(rng));
        SD_STAMP(SD_S_COS);
    } else { // mix_dist{cosine, bvh_mix} (:381-393)
        const uint32_t pick = rng.below(2);
        if (pick == 0) {
            dir = norm(ii.normal + sphere_uniform(rng));
            SD_STAMP(SD_S_COS);
        } else { // bvh_mix_dist::sample :353-361 + triangle_dist::sample :225-239
            const uint32_t id = rng.below(S.lights.n_tris);
            const float4 *lp = LT.tris + 3u * id;
            const float4 l0 = lp[0], l1 = lp[1], l2 = lp[2];
            float u = uniform_real(rng, 0, 1);
            float v = uniform_real(rng, 0, 1);
            if (u + v > 1) {
                u = 1 - u;
                v = 1 - v;
            }
            V3 p = mk(l0.x, l0.y, l0.z) + mk(l0.w, l1.x, l1.y) * v + mk(l1.z, l1.w, l2.x) * u; // a + v' * v + u' * u (DevTri: a, v, u)
            dir = norm(p - pos);
            SD_STAMP(SD_S_LIGHT);
        }
    }
    if (isnan_f(dir.x) || isnan_f(dir.y) || isnan_f(dir.z)) { // :569-571
        out.terminal = true;
        out.term = ii.emission;
        return out;
    }
    const float VNDF_p = vndf_pdf(vr, rd, ii.shading_normal, dir);
    SD_STAMP(SD_VNDF_PDF);
    float MIS_p;
    const float cos_p = rmax(dot(ii.normal, dir) / PI_F, 0.0f); // cosine_dist::pdf :123-128
    if (!has_lights) {
        MIS_p = cos_p;
    } else { // mix_dist::pdf :395-407
        float r = 0;
        r += cos_p;
        r += lights_pdf<STATS>(S, LT, pos, dir, stk, st);
        MIS_p = r / 2.0f;
    }
    const float p = VNDF_FACTOR * VNDF_p + (1 - VNDF_FACTOR) * MIS_p;
    if (p < EPS) { // :576-578
        out.terminal = true;
        out.term = ii.emission;
        return out;
    }
    const V3 scl = pbr_brdf(rd, dir, ii) / p * rmax(0.0f, dot(dir, ii.shading_normal));
    if (len2(scl) == 0.0f) { // :584-586
        out.terminal = true;
        out.term = ii.emission;
        return out;
    }
    out.push = true;
    out.emission = ii.emission;
    out.scl = scl;
    out.nro = pos;
    out.nrd = dir;
    return out;
}

} // namespace

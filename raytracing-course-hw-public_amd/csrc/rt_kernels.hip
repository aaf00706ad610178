// rt_kernels.hip — hand-written gfx950 kernels for the per-pixel Monte Carlo render loop.
//
// What moves to the device (reference file:line):
//   render_pixel / trace_ray / shade / cast_ray / gen_ray        src/raytracer.h:527-627
//   sampling distributions + BRDF                                src/raytracer.h:79-343, 350-432
//   BVH::intersect_ray (ordered closest hit)                     src/bvh.h:195-235
//   BVH::foreach_intersection (all hits, light pdf)              src/bvh.h:237-260
//   intersect(ray, triangle) / intersect(ray, aabb)              src/bvh.h:36-65, 137-152
//   to_intersection_info + Texture::sample + material::*_at      src/bvh.h:80-121, src/geometry.h:545-630
//
// Execution model (MI355X-first, not a translation of the std::thread pool of raytracer.h:636-665):
//   * persistent wavefronts; each LANE owns one work item (a pixel; a 256-pixel span in reference-RNG parity
//     mode) and runs its samples in order, so the per-pixel float sum has the reference's order (raytracer.h:621-626);
//   * idle lanes are refilled from a global ticket with a wave ballot + prefix count (one atomic per wave refill),
//     so live paths stay dense in the wave as paths/pixels terminate at different bounces;
//   * the recursion of trace_ray <-> shade is unrolled into one loop iteration per bounce with a per-lane
//     (emission, scale) stack that is folded back-to-front, preserving the Horner order of raytracer.h:588-590;
//   * the recursive BVH descent becomes an explicit per-lane stack of deferred far siblings. Each entry keeps the
//     sibling's entry distance and the enclosing subtree's local best, because the reference prunes a far child
//     against the NEAR SUBTREE's local best only (bvh.h:220-223), never against a global best.
// Arithmetic contract: IEEE binary32, correctly rounded / and sqrt, no FMA contraction (-ffp-contract=off),
// std::min/std::max operand order reproduced by explicit selects (fminf/fmaxf would drop NaNs differently).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_abi.h"
#include "../../include/rt_devspec.h"
#include "rt_device_types.h"
#include "rt_kernels.h"

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
// intersect(ray, aabb, min_dst) bvh.h:137-152. Division is IEEE; min/max keep std::min/max operand order; the
// component reductions follow std::max_element / std::min_element (first extremum, geometry.h:42-50).
DEV bool box_hit(const float *bmin, const float *bmax, V3 o, V3 d, float min_dst, float &dist) {
    V3 i1 = (ld3(bmin) - o) / d;
    V3 i2 = (ld3(bmax) - o) / d;
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

// intersect_ray_triangle + intersect(ray, triangle, min_dst) bvh.h:36-65 (Cramer; xs = (b, c, t)).
// det(c1,c2,c3) = dot(c1, crs(c2,c3)) (geometry.h:26-29); crs(u, -d) is shared by two determinants.
DEV bool tri_hit(const DevTri &tr, V3 o, V3 d, float min_dst, V3 &xs_out) {
    V3 av = ld3(tr.v), au = ld3(tr.u);
    V3 at = -d;
    V3 y = o - ld3(tr.a);
    V3 c_ut = crs(au, at);
    float den = dot(av, c_ut);
    V3 xs = V3{dot(y, c_ut), dot(av, crs(y, at)), dot(av, crs(au, y))} / den;
    if (xs.x >= 0 && xs.y >= 0 && xs.x + xs.y <= 1 && xs.z >= min_dst) {
        xs_out = xs;
        return true;
    }
    return false;
}

DEV DevNode load_node(const DevNode *nodes, uint32_t idx) {
    DevNode n;
    const float4 *p = reinterpret_cast<const float4 *>(nodes + idx);
    float4 *q = reinterpret_cast<float4 *>(&n);
    q[0] = p[0];
    q[1] = p[1];
    q[2] = p[2];
    q[3] = p[3];
    return n;
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

// BVH::intersect_ray (bvh.h:170-180, 195-235) with an explicit stack.
//   frame = {far child ref, far entry distance d_far, local best of the ENCLOSING subtree at push time}
//   t_loc = local best t of the subtree being traversed (NaN = no hit yet; fminf ignores NaN operands).
// On pop the far sibling is visited iff the near subtree found nothing or found t > d_far (bvh.h:221), then the
// near result is merged into the enclosing subtree's local best. The global best uses the reference's strict
// "replace iff existing t > new t" rule (bvh.h:132) in DFS order, which equals the nested update_intersection calls.
template <bool STATS>
DEV Hit closest_hit(const DevBvh &bvh, V3 o, V3 d, float min_dst, uint32_t *stk_ref, float *stk_d, float *stk_loc, LaneStats<STATS> &st) {
    Hit best{RT_NONE, 0.f, 0.f, 0.f};
    if (bvh.root == RT_NONE || bvh.n_tris == 0)
        return best;
    constexpr uint32_t DONE = 0xFFFFFFFEu, POP = 0xFFFFFFFDu;
    uint32_t cur = bvh.root;
    int sp = 0;
    float t_loc = RT_NAN;
    while (cur != DONE) {
        if (!(cur & RT_LEAF_FLAG)) {
            st.node();
            st.box(2);
            const DevNode n = load_node(bvh.nodes, cur);
            float dl, dr;
            bool hl = box_hit(n.lmin, n.lmax, o, d, min_dst, dl);
            bool hr = box_hit(n.rmin, n.rmax, o, d, min_dst, dr);
            if (hl && hr) {
                uint32_t near = n.left, far = n.right;
                float dfar = dr;
                if (dl > dr) { // bvh.h:216 (ties keep left first)
                    near = n.right;
                    far = n.left;
                    dfar = dl;
                }
                stk_ref[sp] = far;
                stk_d[sp] = dfar;
                stk_loc[sp] = t_loc;
                ++sp;
                t_loc = RT_NAN;
                cur = near;
            } else if (hl) {
                cur = n.left;
            } else if (hr) {
                cur = n.right;
            } else {
                cur = POP;
            }
        } else {
            st.node();
            uint32_t k = cur & ~RT_LEAF_FLAG;
            uint32_t last;
            do {
                const DevTri tr = load_tri(bvh.tris, k);
                st.tri();
                V3 xs;
                if (tri_hit(tr, o, d, min_dst, xs)) {
                    if (best.k == RT_NONE || best.t > xs.z) {
                        best.k = k;
                        best.b = xs.x;
                        best.c = xs.y;
                        best.t = xs.z;
                    }
                    t_loc = fminf(t_loc, xs.z);
                }
                last = tr.flags & 1u;
                ++k;
            } while (!last);
            cur = POP;
        }
        while (cur == POP) {
            if (sp == 0) {
                cur = DONE;
                break;
            }
            --sp;
            const float t_near = t_loc;
            const float dfar = stk_d[sp];
            t_loc = fminf(stk_loc[sp], t_near);
            if (!(t_near <= dfar)) // !has || t_near > d_far (bvh.h:221)
                cur = stk_ref[sp];
        }
    }
    return best;
}

// bvh_mix_dist::pdf (raytracer.h:363-375) = BVH::foreach_intersection (bvh.h:237-260) over the light BVH summing
// triangle_dist::pdf_at (raytracer.h:255-261) in DFS order (node objects, left subtree, right subtree).
template <bool STATS>
DEV float lights_pdf(const DevScene &S, V3 x, V3 d, uint32_t *stk_ref, LaneStats<STATS> &st) {
    const DevBvh &bvh = S.lights;
    st.lq();
    float res = 0;
    if (bvh.root != RT_NONE && bvh.n_tris != 0) {
        constexpr uint32_t DONE = 0xFFFFFFFEu, POP = 0xFFFFFFFDu;
        uint32_t cur = bvh.root;
        int sp = 0;
        while (cur != DONE) {
            if (!(cur & RT_LEAF_FLAG)) {
                st.lnode();
                st.lbox(2);
                const DevNode n = load_node(bvh.nodes, cur);
                float dl, dr;
                bool hl = box_hit(n.lmin, n.lmax, x, d, EPS, dl);
                bool hr = box_hit(n.rmin, n.rmax, x, d, EPS, dr);
                if (hl && hr) {
                    stk_ref[sp++] = n.right;
                    cur = n.left;
                } else if (hl) {
                    cur = n.left;
                } else if (hr) {
                    cur = n.right;
                } else {
                    cur = POP;
                }
            } else {
                st.lnode();
                uint32_t k = cur & ~RT_LEAF_FLAG;
                uint32_t last;
                do {
                    const DevTri tr = load_tri(bvh.tris, k);
                    st.ltri();
                    V3 xs;
                    if (tri_hit(tr, x, d, EPS, xs)) {
                        st.lhit();
                        const float4 aux = *reinterpret_cast<const float4 *>(S.light_aux + k);
                        V3 y = x + d * xs.z;           // ray.at(t)
                        V3 dir = norm(y - x);          // raytracer.h:259
                        float mult = len2(x - y) / __builtin_fabsf(dot(dir, mk(aux.x, aux.y, aux.z))); // :79-84
                        res += mult / aux.w;
                    }
                    last = tr.flags & 1u;
                    ++k;
                } while (!last);
                cur = POP;
            }
            if (cur == POP)
                cur = sp ? stk_ref[--sp] : DONE;
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
    const DevTexture T = S.textures[tex];
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
    const int last = (int)T.count - 1;
    int i00 = px + py * w;
    int i01 = px + mod_inc(py, h) * w;
    int i10 = mod_inc(px, w) + py * w;
    int i11 = mod_inc(px, w) + mod_inc(py, h) * w;
    // memory-safety clamp only: the reference indexes out of bounds when wrap_repeat rounds up to 1.0f
    i00 = min(max(i00, 0), last);
    i01 = min(max(i01, 0), last);
    i10 = min(max(i10, 0), last);
    i11 = min(max(i11, 0), last);
    const uint32_t *pool = S.texels + T.offset;
    uint32_t q00 = pool[i00], q01 = pool[i01], q10 = pool[i10], q11 = pool[i11];
    st.texels(4);
    const float *rgb = gamma ? s_gam : s_lin;
    auto dec = [&](uint32_t p) { return C4{rgb[p & 255u], rgb[(p >> 8) & 255u], rgb[(p >> 16) & 255u], s_lin[p >> 24]}; };
    C4 p00 = dec(q00), p01 = dec(q01), p10 = dec(q10), p11 = dec(q11);
    return (1 - dx) * ((1 - dy) * p00 + dy * p01) + dx * ((1 - dy) * p10 + dy * p11);
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
DEV Surf make_surf(const DevScene &S, const Hit &h, V3 rd, const float *s_lin, const float *s_gam, LaneStats<STATS> &st) {
    DevAttr at;
    {
        const float4 *p = reinterpret_cast<const float4 *>(S.attrs + h.k);
        float4 *q = reinterpret_cast<float4 *>(&at);
#pragma unroll
        for (int i = 0; i < 7; ++i)
            q[i] = p[i];
    }
    DevMaterial m;
    {
        const float4 *p = reinterpret_cast<const float4 *>(S.materials + at.material);
        float4 *q = reinterpret_cast<float4 *>(&m);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            q[i] = p[i];
    }
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
    C4 nt = tex_sample(S, m.normal_tex, TEX_DEFAULT_NORMAL_UP, tu, tv, false, s_lin, s_gam, st); // sample_normal geometry.h:577-582
    V3 normal_loc = norm(mk(nt.r, nt.g, nt.b) * 2 - 1);
    V3 shading = norm(transform3(normal_loc, tangent, bitangent, smooth));
    C4 mr = tex_sample(S, m.mr_tex, TEX_DEFAULT_WHITE, tu, tv, false, s_lin, s_gam, st); // geometry.h:623-626
    C4 ct = tex_sample(S, m.color_tex, TEX_DEFAULT_WHITE, tu, tv, true, s_lin, s_gam, st); // :615-617
    C4 et = tex_sample(S, m.emissive_tex, TEX_DEFAULT_WHITE, tu, tv, true, s_lin, s_gam, st); // :619-621
    st.shaded();
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
    rt_sincos(phi, &s, &c);
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
    rt_sincos(phi, &sn, &cs);
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

// ---------------------------------------------------------------------------------------------- render kernel
struct Item {
    uint32_t pix, pix_end, seed;
};
DEV Item map_item(const RenderLaunch &L, uint32_t item) {
    const uint32_t n_pix = L.width * L.height;
    const uint32_t unit = (L.rng_mode == RT_RNG_REFERENCE) ? RT_SPAN : 1u;
    uint32_t local_block = item / L.items_per_block;
    uint32_t within = item - local_block * L.items_per_block;
    uint32_t global_block = local_block * L.shard_count + L.shard_index;
    uint32_t first_unit = global_block * L.items_per_block + within; // global pixel (device) / span (reference) index
    Item it;
    it.pix = first_unit * unit;
    uint32_t end = it.pix + unit;
    it.pix_end = end < n_pix ? end : n_pix;
    it.seed = first_unit;
    return it;
}

template <int MODE, bool STATS>
__global__ __launch_bounds__(256) void render_kernel(const DevScene S, const RenderLaunch L) {
    __shared__ float s_lin[256];
    __shared__ float s_gam[256];
    s_lin[threadIdx.x] = S.lut_linear[threadIdx.x];
    s_gam[threadIdx.x] = S.lut_gamma[threadIdx.x];
    __syncthreads();

    LaneStats<STATS> st;
    Rng<MODE> rng;
    uint32_t stk_ref[RT_MAX_STACK];
    float stk_d[RT_MAX_STACK];
    float stk_loc[RT_MAX_STACK];
    float fold_e[RT_MAX_RAY_DEPTH * 3];
    float fold_s[RT_MAX_RAY_DEPTH * 3];

    const V3 cam_pos = ld3(S.cam_pos), cam_right = ld3(S.cam_right), cam_up = ld3(S.cam_up), cam_fwd = ld3(S.cam_fwd);
    const bool has_lights = S.lights.n_tris != 0; // raytracer.h:449-453

    bool have_item = false, exhausted = false, path_live = false;
    uint32_t pix = 0, pix_end = 0, s = 0, depth_left = 0, nb = 0;
    V3 acc{0, 0, 0}, ro{0, 0, 0}, rd{0, 0, 1};

    for (;;) {
        // ---- refill idle lanes: ballot + prefix count, one ticket atomic per wave
        if (!have_item && !exhausted) {
            const unsigned long long m = __ballot(1);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            uint32_t base = 0;
            if (rank == 0)
                base = atomicAdd(L.counter, (uint32_t)__popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            const uint32_t item = base + rank;
            if (item < L.n_items) {
                Item it = map_item(L, item);
                pix = it.pix;
                pix_end = it.pix_end;
                s = 0;
                acc = mk(0, 0, 0);
                have_item = pix < pix_end;
                path_live = false;
                if constexpr (MODE == RT_RNG_REFERENCE)
                    rt_minstd_seed(&rng.g, it.seed); // RaytracerThreadContext(ctx, span) :648
            } else {
                exhausted = true;
            }
        }
        if (__ballot(have_item) == 0ull)
            break;
        if (!have_item)
            continue;

        if (!path_live) { // render_pixel loop head :621-622 + gen_ray :527-538
            if constexpr (MODE == RT_RNG_DEVICE)
                rt_xoshiro_seed(&rng.g, L.seed, pix, s);
            const uint32_t x = pix % L.width, y = pix / L.width;
            float ox = uniform_real(rng, 0.0f, 1.0f);
            float oy = uniform_real(rng, 0.0f, 1.0f);
            float sx = (2 * ((float)(int)x + ox) / (float)L.width - 1) * L.tan_x;
            float sy = (2 * ((float)(int)y + oy) / (float)L.height - 1) * L.tan_y;
            rd = norm(sx * cam_right - sy * cam_up + 1.0f * cam_fwd);
            ro = cam_pos;
            depth_left = S.ray_depth;
            nb = 0;
            path_live = true;
        }

        // ---- one trace_ray level (:593-605)
        bool terminal = false;
        V3 term{0, 0, 0};
        if (depth_left == 0) {
            terminal = true;
        } else {
            st.cast();
            const Hit h = closest_hit<STATS>(S.scene, ro, rd, EPS, stk_ref, stk_d, stk_loc, st);
            if (h.k == RT_NONE) {
                terminal = true;
                term = ld3(S.bg) * mk(1, 1, 1); // Scene::bg_at with the 1x1 white bg (scene.h:83-89)
            } else {
                depth_left -= 1; // shade(..., max_depth - 1)
                const Surf ii = make_surf<STATS>(S, h, rd, s_lin, s_gam, st);
                const V3 pos = ro + rd * h.t; // ray.at(t)
                if (!(uniform_real(rng, 0.0f, 1.0f) <= ii.color.a)) { // !coin(alpha) :559-561
                    ro = pos;
                } else {
                    const float vr = pow2(rmax(ii.roughness, MIN_ROUGHNESS)); // :563-564
                    V3 dir;
                    if (uniform_real(rng, 0.0f, 1.0f) <= VNDF_FACTOR) { // :565
                        dir = vndf_sample(rng, vr, rd, ii.shading_normal);
                    } else if (!has_lights) { // dir_dist = cosine_dist (:449)
                        dir = norm(ii.normal + sphere_uniform(rng));
                    } else { // mix_dist{cosine, bvh_mix} (:381-393)
                        const uint32_t pick = rng.below(2);
                        if (pick == 0) {
                            dir = norm(ii.normal + sphere_uniform(rng));
                        } else { // bvh_mix_dist::sample :353-361 + triangle_dist::sample :225-239
                            const uint32_t id = rng.below(S.lights.n_tris);
                            const DevTri lt = load_tri(S.lights.tris, id);
                            float u = uniform_real(rng, 0, 1);
                            float v = uniform_real(rng, 0, 1);
                            if (u + v > 1) {
                                u = 1 - u;
                                v = 1 - v;
                            }
                            V3 p = ld3(lt.a) + ld3(lt.v) * v + ld3(lt.u) * u;
                            dir = norm(p - pos);
                        }
                    }
                    if (isnan_f(dir.x) || isnan_f(dir.y) || isnan_f(dir.z)) { // :569-571
                        terminal = true;
                        term = ii.emission;
                    } else {
                        const float VNDF_p = vndf_pdf(vr, rd, ii.shading_normal, dir);
                        float MIS_p;
                        const float cos_p = rmax(dot(ii.normal, dir) / PI_F, 0.0f); // cosine_dist::pdf :123-128
                        if (!has_lights) {
                            MIS_p = cos_p;
                        } else { // mix_dist::pdf :395-407
                            float r = 0;
                            r += cos_p;
                            r += lights_pdf<STATS>(S, pos, dir, stk_ref, st);
                            MIS_p = r / 2.0f;
                        }
                        const float p = VNDF_FACTOR * VNDF_p + (1 - VNDF_FACTOR) * MIS_p;
                        if (p < EPS) { // :576-578
                            terminal = true;
                            term = ii.emission;
                        } else {
                            const V3 scl = pbr_brdf(rd, dir, ii) / p * rmax(0.0f, dot(dir, ii.shading_normal));
                            if (len2(scl) == 0.0f) { // :584-586
                                terminal = true;
                                term = ii.emission;
                            } else { // emission + trace_ray(...) * scl  (:588-590) deferred to the fold below
                                fold_e[3 * nb + 0] = ii.emission.x;
                                fold_e[3 * nb + 1] = ii.emission.y;
                                fold_e[3 * nb + 2] = ii.emission.z;
                                fold_s[3 * nb + 0] = scl.x;
                                fold_s[3 * nb + 1] = scl.y;
                                fold_s[3 * nb + 2] = scl.z;
                                ++nb;
                                ro = pos;
                                rd = dir;
                            }
                        }
                    }
                }
            }
        }

        if (terminal) {
            V3 res = term;
            while (nb > 0) { // unwind shade() frames: emission + clr, clr = inner * scl
                --nb;
                V3 clr = res * mk(fold_s[3 * nb], fold_s[3 * nb + 1], fold_s[3 * nb + 2]);
                res = mk(fold_e[3 * nb], fold_e[3 * nb + 1], fold_e[3 * nb + 2]) + clr;
            }
            if (isnan_f(res.x)) // sanitize_nans :607-616
                res.x = 0;
            if (isnan_f(res.y))
                res.y = 0;
            if (isnan_f(res.z))
                res.z = 0;
            acc = acc + res;
            st.sample();
            path_live = false;
            ++s;
            if (s == L.samples) { // return res / samples :626
                const V3 out = acc / (float)L.samples;
                float *dst = L.fb + 3ull * pix;
                dst[0] = out.x;
                dst[1] = out.y;
                dst[2] = out.z;
                ++pix;
                s = 0;
                acc = mk(0, 0, 0);
                if (pix == pix_end)
                    have_item = false;
            }
        }
    }
    st.flush(L.stats);
}

// ---------------------------------------------------------------------------------------------- probe kernels
__global__ __launch_bounds__(256) void cast_kernel(const DevScene S, const float *rays, uint32_t n, uint32_t *prim_out, float *bct_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    uint32_t stk_ref[RT_MAX_STACK];
    float stk_d[RT_MAX_STACK];
    float stk_loc[RT_MAX_STACK];
    LaneStats<false> st;
    V3 o = ld3(rays + 6ull * i), d = ld3(rays + 6ull * i + 3);
    Hit h = closest_hit<false>(S.scene, o, d, EPS, stk_ref, stk_d, stk_loc, st);
    if (h.k == RT_NONE) {
        prim_out[i] = RT_NONE;
        bct_out[3ull * i] = bct_out[3ull * i + 1] = bct_out[3ull * i + 2] = 0.0f;
    } else {
        prim_out[i] = S.scene.tris[h.k].prim;
        bct_out[3ull * i] = h.b;
        bct_out[3ull * i + 1] = h.c;
        bct_out[3ull * i + 2] = h.t;
    }
}

__global__ __launch_bounds__(256) void light_pdf_kernel(const DevScene S, const float *rays, uint32_t n, float *pdf_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    uint32_t stk_ref[RT_MAX_STACK];
    LaneStats<false> st;
    V3 o = ld3(rays + 6ull * i), d = ld3(rays + 6ull * i + 3);
    pdf_out[i] = S.lights.n_tris ? lights_pdf<false>(S, o, d, stk_ref, st) : 0.0f;
}

} // namespace

namespace rt {

hipError_t launch_render(const DevScene &S, const RenderLaunch &L, bool stats, int blocks, hipStream_t stream) {
    dim3 grid(blocks), block(256);
    if (L.rng_mode == RT_RNG_REFERENCE) {
        if (stats)
            hipLaunchKernelGGL((render_kernel<RT_RNG_REFERENCE, true>), grid, block, 0, stream, S, L);
        else
            hipLaunchKernelGGL((render_kernel<RT_RNG_REFERENCE, false>), grid, block, 0, stream, S, L);
    } else {
        if (stats)
            hipLaunchKernelGGL((render_kernel<RT_RNG_DEVICE, true>), grid, block, 0, stream, S, L);
        else
            hipLaunchKernelGGL((render_kernel<RT_RNG_DEVICE, false>), grid, block, 0, stream, S, L);
    }
    return hipGetLastError();
}

hipError_t launch_cast(const DevScene &S, const float *rays, uint32_t n, uint32_t *prim, float *bct, hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(cast_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, S, rays, n, prim, bct);
    return hipGetLastError();
}

hipError_t launch_light_pdf(const DevScene &S, const float *rays, uint32_t n, float *pdf, hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    hipLaunchKernelGGL(light_pdf_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, S, rays, n, pdf);
    return hipGetLastError();
}

} // namespace rt

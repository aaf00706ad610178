// rt_wide.hip — closest hit through the 8-wide BVH with quantised child boxes (WideNode, rt_device_types.h): the production
// traversal of the wavefront pipeline (scenes created with RT_BUILD_WIDE). Same queue protocol as wf_extend (rt_wavefront.hip):
// persistent wavefronts, lanes refilled from the ray queue as their traversals end, hits stored at the ray's queue position.
//
// What replaces the reference's recursion (BVH::intersect_ray, src/bvh.h:195-235):
//   * a node visit tests EIGHT child boxes from one 64-byte record, 64-byte aligned (4 vector-L1 accesses per lane, never more than one
//     128-byte line; the binary node costs 4 for two boxes and the L1 access rate is what bounds these kernels, profiles/r02_l1_roof.txt,
//     profiles/r04_variants.txt). The builders emit 80-byte WideNode records; rt_wide_pack.hip re-encodes them (origin as 3 x 20 bits on a
//     scene grid, 4-bit exponents, slot states) and lays nodes and triangle records out in ONE array of 16-byte units: a node's inner
//     children, then its leaf triangles, addressed by one base;
//   * every box is culled against the GLOBAL best hit so far (the reference prunes only between siblings, bvh.h:216-223);
//   * no distance sort: a node's slots are laid out by octant (wide_build.cpp), so the hit slots are visited front to back by
//     taking them in the order of decreasing (slot ^ oct ^ 7), oct = the ray's direction signs; the pending rest of a node is
//     ONE 8-byte stack frame {first child index, pending slots | inner-slot mask}, whatever the number of hit children;
//   * triangles are tested exactly as everywhere else (tri_hit, rt_device_lib.h: the reference's Cramer solve, bvh.h:36-65), in
//     wave-wide batches like wf_extend's leaf batches, so a hit's (b, c, t) are bit-equal to the reference's for that triangle.
// The boxes are NOT the reference's (8-bit conservative supersets, slab test by fused multiply-adds with an explicit error
// margin instead of IEEE division): a ray can only see MORE boxes than exact arithmetic would allow, never fewer, so every
// triangle the reference finds is found; on exact ties another triangle index may win. Checked against the CPU oracle in
// tests/test_gpu_production.py (t bit-equal on every ray, index differences counted and confined to ties).
#include "rt_device_lib.h"
#include "rt_kernels.h"

namespace {

#ifndef RT_WIDE_WAVES_PER_SIMD
#define RT_WIDE_WAVES_PER_SIMD 5
#endif
#ifndef RT_WIDE_LDS_DEPTH
#define RT_WIDE_LDS_DEPTH 8 /* stacked node groups kept in LDS per lane (8 B each); deeper ones spill to the overflow workspace */
#endif
#ifndef RT_WIDE_CHUNK
#define RT_WIDE_CHUNK 64u
#endif
#ifndef RT_WIDE_REFILL_MIN
#define RT_WIDE_REFILL_MIN 16
#endif
#ifndef RT_WIDE_TRI_MIN
#define RT_WIDE_TRI_MIN 20 /* run a triangle batch once this many lanes wait with pending triangles */
#endif
#define RT_WIDE_COOP_MAX 8u /* triangles one lane contributes to a batch; more stay pending for the next one */

// What a node visit needs of the ray, computed ONCE where the ray is taken from the queue (round 4: the packed node's origin is an index m on the scene
// grid, so the ray-dependent half of every plane equation can leave the node step): on axis a the parameter of the node's plane q is
//     t = q * ad + m * gi + c,     ad = cell / d = ids * 2^e4,   gi = g / d,   c = (grid base - o) / d,
// and the conservative margin is folded into c: `lo` for entry planes, `hi` for exit planes. Margin: every magnitude in that sum is below
// B = (|base - o| + 2^20 g) |1/d|, five roundings of <= 2^-24 of such magnitudes are involved -> 2^-19 B on either side (about 1e-4 of a
// parameter unit on the bench scenes, against node extents of 0.25 and more): a box can only be entered earlier / left later than in exact arithmetic.
struct WRay {
    V3 lo, hi; // c -/+ margin
    V3 gi;     // g / d
    V3 ids;    // (1/d) * 2^e_base: a node's ad is this with the node's 4-bit exponent added to the exponent field
};
struct WTrav {
    V3 o, d;           // for the triangle tests
    WRay w;
    uint32_t oct_inv;  // 7 ^ direction-sign octant
    uint32_t gx, gy;   // current node group: first inner child of the node | pending slots by priority (bits 31..24), inner-slot mask (bits 7..0)
    uint32_t top_x, top_y; // newest stacked group, cached in registers
    int sp;
    uint32_t tbase, tm, tall; // pending triangles of the node just tested: DevTri base, hit bits, all bits (for the compact index)
    Hit best;          // best.t = +inf until the first hit
    bool done;
};

// a direction component of (almost) zero would make 0 * inf = NaN in the slab terms: clamp |1/d| to 2^60. The slab then spans
// (-huge, +huge) for an origin inside it, is empty for one outside, and the boundary case counts as inside (conservative). A component
// beyond 2^40 (1/d below 2^-40; no unit-length direction has one) is clamped too, so that the exponent arithmetic below stays among normal floats
// (rt_scene.cpp bounds the scene's cell exponents to match).
DEV float wide_clamp_idir(float d, float r) {
    const float big = 1152921504606846976.0f, small = 8.673617379884035e-19f; // 2^60, 2^-60
    const float a = __builtin_fabsf(d);
    return a < small ? __builtin_copysignf(big, d) : (a > 1099511627776.0f ? __builtin_copysignf(9.094947017729282e-13f, d) : r); // 2^40 -> 2^-40
}
DEV WRay wide_ray(const WideGrid &G, V3 o, V3 idir) {
    WRay w;
    const float scale = __uint_as_float((uint32_t)(G.e_base + 127) << 23); // 2^e_base
    const float span = 1048576.0f * G.g;                                    // 2^20 g: beyond every m * g
    const V3 c0 = mk(G.base[0] - o.x, G.base[1] - o.y, G.base[2] - o.z);
    const V3 c = c0 * idir;
    const V3 e = mk((__builtin_fabsf(c0.x) + span) * __builtin_fabsf(idir.x), (__builtin_fabsf(c0.y) + span) * __builtin_fabsf(idir.y),
                    (__builtin_fabsf(c0.z) + span) * __builtin_fabsf(idir.z)) * 1.9073486328125e-06f; // 2^-19
    w.lo = c - e;
    w.hi = c + e;
    w.gi = idir * G.g;
    w.ids = idir * scale;
    return w;
}

DEV void wide_init(WTrav &T, const DevBvh &bvh, V3 o, V3 d, V3 r) {
    T.o = o;
    T.d = d;
    const V3 idir = mk(wide_clamp_idir(d.x, r.x), wide_clamp_idir(d.y, r.y), wide_clamp_idir(d.z, r.z));
    T.w = wide_ray(bvh.grid, o, idir);
    const uint32_t oct = (idir.x < 0.0f ? 1u : 0u) | (idir.y < 0.0f ? 2u : 0u) | (idir.z < 0.0f ? 4u : 0u);
    T.oct_inv = 7u ^ oct;
    T.gx = 0u;
    T.gy = bvh.n_wide != 0u ? 0x80000000u : 0u; // "slot oct of a node whose children start at record 0": the root, whatever oct is (imask 0)
    T.top_x = T.top_y = 0u;
    T.sp = 0;
    T.tbase = T.tm = T.tall = 0u;
    T.best = Hit{RT_NONE, 0.f, 0.f, RT_INF};
    T.done = false;
}

DEV float ub(uint32_t w, int k) { return (float)((w >> (8 * k)) & 255u); } // v_cvt_f32_ubyteK

// The packed node's header (piece 0, rt_device_types.h): origin index m on the scene grid, 4-bit cell exponents, slot states, base.
struct WideHdr {
    uint32_t mx, my, mz;    // origin = grid base + m * g
    uint32_t ex, ey, ez;    // the 4-bit cell exponents, already at the position of a float's exponent field (<< 23)
    uint32_t base, imask, tri_mask; // first inner child (16-byte unit index), inner-slot mask, leaf triangles (3 bits per slot, as WideNode::tri_mask)
};
DEV WideHdr wide_decode(const uint4 h) {
    WideHdr H;
    H.mx = h.z & 0xFFFFFu, H.my = __builtin_amdgcn_alignbit(h.w, h.z, 20) & 0xFFFFFu, H.mz = (h.w >> 8) & 0xFFFFFu;
    H.ex = (h.y >> 1) & 0x07800000u, H.ey = (h.y >> 5) & 0x07800000u, H.ez = (h.w >> 5) & 0x07800000u;
    const uint32_t state = h.y & 0xFFFFFFu;
    uint32_t t = (state >> 2) & ~state & 0x249249u; // bit 3s: slot s holds the pattern 100 = an inner node
    H.tri_mask = state & ~(t << 2);
    t = (t | (t >> 2)) & 0x0C30C3u; // every third bit -> consecutive bits
    t = (t | (t >> 4)) & 0x00F00Fu;
    H.imask = (t | (t >> 8)) & 0xFFu;
    H.base = h.x;
    return H;
}

// The eight slab tests of one node record against one ray: bit i of the result = the ray meets slot i's box within [EPS, tlim].
// Plane q of axis a: t = q * ad + (m * gi + lo / hi), see WRay: two FMAs per plane pair and axis for the node-dependent constant, one FMA per plane.
DEV uint32_t wide_test8(const WideHdr &H, const uint4 n2, const uint4 n3, const uint4 n4, const WRay &w, float tlim) {
    const float adx = __uint_as_float(__float_as_uint(w.ids.x) + H.ex), ady = __uint_as_float(__float_as_uint(w.ids.y) + H.ey),
                adz = __uint_as_float(__float_as_uint(w.ids.z) + H.ez);
    const float fx = (float)H.mx, fy = (float)H.my, fz = (float)H.mz;
    const float bx0 = __builtin_fmaf(fx, w.gi.x, w.lo.x), bx1 = __builtin_fmaf(fx, w.gi.x, w.hi.x), by0 = __builtin_fmaf(fy, w.gi.y, w.lo.y),
                by1 = __builtin_fmaf(fy, w.gi.y, w.hi.y), bz0 = __builtin_fmaf(fz, w.gi.z, w.lo.z), bz1 = __builtin_fmaf(fz, w.gi.z, w.hi.z);
    // near / far planes per axis by the direction sign: words {slots 0..3, slots 4..7}. Selected with a per-lane BIT mask (the sign of 1/d smeared
    // over the word: one v_bfi_b32 per select) instead of `cond ? a : b` (one compare and four v_cndmask on vcc per axis): equally fast
    // (profiles/r04_variants.txt item 3: only DEPENDENT back-to-back vcc selects are slow on gfx950), three compares fewer.
    uint32_t mx = (uint32_t)((int32_t)__float_as_uint(w.ids.x) >> 31), my = (uint32_t)((int32_t)__float_as_uint(w.ids.y) >> 31),
             mz = (uint32_t)((int32_t)__float_as_uint(w.ids.z) >> 31);
    asm volatile("" : "+v"(mx), "+v"(my), "+v"(mz)); // opaque: otherwise the masks are recognised as sign tests and the selects come back as v_cndmask on vcc
    auto pick = [](uint32_t m, uint32_t if_set, uint32_t if_clear) { return (if_set & m) | (if_clear & ~m); };
    const uint32_t xn0 = pick(mx, n3.z, n2.x), xn1 = pick(mx, n3.w, n2.y), xf0 = pick(mx, n2.x, n3.z), xf1 = pick(mx, n2.y, n3.w);
    const uint32_t yn0 = pick(my, n4.x, n2.z), yn1 = pick(my, n4.y, n2.w), yf0 = pick(my, n2.z, n4.x), yf1 = pick(my, n2.w, n4.y);
    const uint32_t zn0 = pick(mz, n4.z, n3.x), zn1 = pick(mz, n4.w, n3.y), zf0 = pick(mz, n3.x, n4.z), zf1 = pick(mz, n3.y, n4.w);
    uint32_t h = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = i & 3;
        const uint32_t wxn = i < 4 ? xn0 : xn1, wxf = i < 4 ? xf0 : xf1, wyn = i < 4 ? yn0 : yn1, wyf = i < 4 ? yf0 : yf1, wzn = i < 4 ? zn0 : zn1, wzf = i < 4 ? zf0 : zf1;
        const float tx0 = __builtin_fmaf(ub(wxn, k), adx, bx0), ty0 = __builtin_fmaf(ub(wyn, k), ady, by0), tz0 = __builtin_fmaf(ub(wzn, k), adz, bz0);
        const float tx1 = __builtin_fmaf(ub(wxf, k), adx, bx1), ty1 = __builtin_fmaf(ub(wyf, k), ady, by1), tz1 = __builtin_fmaf(ub(wzf, k), adz, bz1);
        const float tmin = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, EPS));
        const float tmax = fminf(fminf(tx1, ty1), fminf(tz1, tlim));
        h |= tmin <= tmax ? (1u << i) : 0u;
    }
    return h;
}
// inner-slot hits -> priority bits (slot ^ oct_inv): an xor of the bit INDEX = three conditional block swaps of the byte
DEV uint32_t wide_priority(uint32_t r, uint32_t oct_inv) {
    const uint32_t s1 = ((r & 0x55u) << 1) | ((r >> 1) & 0x55u);
    r = (oct_inv & 1u) ? s1 : r;
    const uint32_t s2 = ((r & 0x33u) << 2) | ((r >> 2) & 0x33u);
    r = (oct_inv & 2u) ? s2 : r;
    const uint32_t s4 = ((r & 0x0Fu) << 4) | (r >> 4);
    return (oct_inv & 4u) ? s4 : r;
}
// leaf-slot hits -> their triangles: every bit of the slot mask tripled, then only the triangles that exist
DEV uint32_t wide_leaf_tris(uint32_t l, uint32_t tri_mask) {
    l &= 255u;
    l = (l | (l << 8)) & 0x0000F00Fu;
    l = (l | (l << 4)) & 0x000C30C3u;
    l = (l | (l << 2)) & 0x00249249u;
    return (l * 7u) & tri_mask;
}

// One node visit: take the next child of the current group (pushing the rest back if any), fetch its record and test its
// eight boxes against [EPS, best.t]. Leaves the child's own group in (gx, gy) and its hit triangles in (tbase, tm, tall).
template <bool STATS, class STK> DEV void wide_node_step(WTrav &T, const uint4 *blob, STK &stk, LaneStats<STATS> &st) {
    const uint32_t hits = T.gy;
    const uint32_t bit = 31u - (uint32_t)__clz((int)hits);
    const uint32_t rest = hits ^ (1u << bit);
    const uint32_t slot = (bit - 24u) ^ T.oct_inv;
    const uint32_t idx = T.gx + RT_WIDE_NODE_UNITS * (uint32_t)__popc(hits & 0xFFu & ((1u << slot) - 1u)); // 16-byte unit index of the child's record
    const bool more = (rest >> 24) != 0u;
    if (more & (T.sp > 0))
        stk.push(T.sp - 1, T.top_x, __uint_as_float(T.top_y)); // spill the previous top
    T.top_x = more ? T.gx : T.top_x;
    T.top_y = more ? rest : T.top_y;
    T.sp += more ? 1 : 0;

    const uint4 *p = blob + idx;
    const uint4 n0 = p[0], n2 = p[1], n3 = p[2], n4 = p[3];
    st.node();
    st.box(8);
    const WideHdr H = wide_decode(n0);
    const uint32_t imask = H.imask;
    const uint32_t h = wide_test8(H, n2, n3, n4, T.w, T.best.t);
    T.gx = H.base;
    T.gy = (wide_priority(h & imask, T.oct_inv) << 24) | imask;
    T.tbase = H.base + RT_WIDE_NODE_UNITS * (uint32_t)__popc(imask); // the node's triangle records follow its inner children
    T.tall = H.tri_mask;
    T.tm = wide_leaf_tris(h & ~imask, H.tri_mask);
#ifdef RT_WIDE_DIAG // development census through the (otherwise idle) light counters of the instrumented variant
    if (h == 0u)
        st.lhit(); // a visit that hit none of the eight boxes
    st.lbox((uint32_t)__popc(h & imask)); // inner children hit
    st.ltri();                            // (visits, again: denominator)
    if ((h & ~imask & 255u) != 0u)
        st.lnode(); // visits with at least one leaf slot hit
#endif
}

// Triangle batch, as wf_extend's leaf batch: the pending (ray, triangle) pairs of all waiting lanes are laid out densely over
// the wave; a lane's result is the minimum of (t bits, triangle record) over its pairs: smallest t, lowest record on equal t.
template <bool STATS>
DEV void wide_tri_batch(WTrav &T, const uint4 *blob, bool waiting, uint16_t *s_owner, unsigned long long *s_min, float4 *s_bc, LaneStats<STATS> &st) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t have = waiting ? (uint32_t)__popc(T.tm) : 0u;
    const uint32_t n = have < RT_WIDE_COOP_MAX ? have : RT_WIDE_COOP_MAX;
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) { // exclusive prefix sum of n (0..8) over the wave, one ballot per bit plane
        const unsigned long long m = __ballot((n >> b) & 1u);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        off += below << b;
        total += (uint32_t)__popcll(m) << b;
    }
    // Owner table: position off + t belongs to (lane, compact record offset of the lane's t-th pending triangle). Every
    // waiting lane writes all RT_WIDE_COOP_MAX entries, highest t first, without a per-entry predicate: an entry with t >= n
    // lands on a position of a later lane, whose own store of that position is issued later and wins (see leaf_batch).
    if (waiting) {
        uint32_t m = T.tm;
        uint16_t ent[RT_WIDE_COOP_MAX];
#pragma unroll
        for (int t = 0; t < (int)RT_WIDE_COOP_MAX; ++t) {
            const uint32_t b = (uint32_t)(__ffs((int)m) - 1) & 31u;
            const uint32_t rec = (uint32_t)__popc(T.tall & ((1u << b) - 1u)); // compact index of bit b
            ent[t] = (uint16_t)(lane | (rec << 8));
            m &= m - 1u;
        }
#pragma unroll
        for (int t = (int)RT_WIDE_COOP_MAX - 1; t >= 0; --t) {
            s_owner[off + t] = ent[t];
            asm volatile("" ::: "memory"); // keep the stores in this order
        }
        s_min[lane] = ~0ull;
        T.tm = m; // what did not fit this batch stays pending
    }
    __threadfence_block();
    for (uint32_t q0 = 0; q0 < total; q0 += 64u) { // wave-uniform trip count
        const uint32_t q = q0 + lane;
        const bool valid = q < total;
        const uint32_t ow = valid ? (uint32_t)s_owner[q] : 0u;
        const int src = (int)(ow & 63u);
        const uint32_t kk = (uint32_t)__shfl((int)T.tbase, src) + RT_WIDE_TRI_UNITS * (ow >> 8); // 16-byte unit index of the triangle record
        const V3 o = mk(__shfl(T.o.x, src), __shfl(T.o.y, src), __shfl(T.o.z, src));
        const V3 d = mk(__shfl(T.d.x, src), __shfl(T.d.y, src), __shfl(T.d.z, src));
        if (valid) {
            const float4 *p = reinterpret_cast<const float4 *>(blob + kk);
            const float4 r0 = p[0], r1 = p[1], r2 = p[2];
            st.tri();
            V3 xs;
            if (tri_hit(mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), o, d, EPS, xs)) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(xs.z) << 32) | (unsigned long long)kk;
                atomicMin(&s_min[src], key);
                __threadfence_block();
                if (s_min[src] == key) // this pair leads its ray so far: publish its barycentrics and the record's DevTri / DevAttr index (DevTri::pad)
                    s_bc[src] = make_float4(xs.x, xs.y, r2.w, 0.0f);
            }
        }
    }
    __threadfence_block();
    if (waiting) {
        const unsigned long long key = s_min[lane];
        if (key != ~0ull) {
            const float t = __uint_as_float((uint32_t)(key >> 32));
            const float4 bc = s_bc[lane];
            if (T.best.t > t) { // strict: the first-found triangle keeps an exact tie (update_intersection, bvh.h:132)
                T.best.k = __float_as_uint(bc.z);
                T.best.b = bc.x;
                T.best.c = bc.y;
                T.best.t = t;
            }
        }
    }
}

template <bool STATS> __global__ __launch_bounds__(256, RT_WIDE_WAVES_PER_SIMD) void wf_extend_wide(const DevScene S, const WfLaunch L) {
    __shared__ uint32_t s_stack[STACK_LDS_DWORDS_W(RT_WIDE_LDS_DEPTH, 2)];
    __shared__ uint16_t s_owner_all[4][64 * RT_WIDE_COOP_MAX + RT_WIDE_COOP_MAX]; // + overshoot of the unpredicated owner stores
    __shared__ unsigned long long s_min_all[4][64];
    __shared__ float4 s_bc_all[4][64];
    const uint32_t wave = threadIdx.x >> 6;
    uint16_t *s_owner = s_owner_all[wave];
    unsigned long long *s_min = s_min_all[wave];
    float4 *s_bc = s_bc_all[wave];
    LaneStats<STATS> st;
    RT_DECLARE_RING_STACK_W(stk, RT_WIDE_LDS_DEPTH, 2, s_stack, L.stack_overflow, L.stack_stride);
    const uint32_t n_in = L.counters[WF_CNT_IN];
    const uint4 *blob = reinterpret_cast<const uint4 *>(S.scene.wide);
    WTrav T;
    T.o = T.d = mk(0.f, 0.f, 0.f);
    T.w.lo = T.w.hi = T.w.gi = T.w.ids = mk(0.f, 0.f, 0.f);
    T.oct_inv = 0u;
    T.gx = T.gy = T.top_x = T.top_y = 0u;
    T.sp = 0;
    T.tbase = T.tm = T.tall = 0u;
    T.best = Hit{RT_NONE, 0.f, 0.f, RT_INF};
    T.done = true;
    uint32_t slot = RT_NONE;
    bool exhausted = n_in == 0; // wave-uniform
    uint32_t q_lo = 0, q_hi = 0;
    TicketState tks = ticket_init();
#ifdef RT_DIAG_CYCLES
    // section census of the persistent loop (development build only): wave cycles between s_memtime stamps, trips and lanes per section
    unsigned long long dg_refill = 0, dg_node = 0, dg_tri = 0, dg_pop = 0, dg_nodes_n = 0, dg_nodes_lanes = 0, dg_tri_n = 0, dg_tri_lanes = 0, dg_t = __builtin_amdgcn_s_memtime();
    const unsigned long long dg_start = dg_t;
#define WDG_STAMP(acc)                                                \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        acc += now_ - dg_t;                                           \
        dg_t = now_;                                                  \
    } while (0)
#else
#define WDG_STAMP(acc) do { } while (0)
#endif
    for (;;) {
        const bool idle = T.done;
        const unsigned long long im = __ballot(idle);
        const int n_idle = __popcll(im);
        if (!exhausted && (n_idle >= RT_WIDE_REFILL_MIN || n_idle == (int)__popcll(__ballot(1)))) {
            if (q_lo == q_hi) // a new range of queue positions, one ticket atomic per chunk
                exhausted = !ticket_take(L.counters, n_in, (uint32_t)RT_WIDE_CHUNK, tks, q_lo, q_hi);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
            const uint32_t avail = q_hi - q_lo;
            if (idle && rank < avail) {
                const uint32_t jq = q_lo + rank;
                const uint32_t j = L.order ? L.order[jq] & WF_ORDER_SLOT_MASK : jq;
                const float4 *rq = reinterpret_cast<const float4 *>(L.paths_in + j);
                const float4 r0 = rq[0], r1 = rq[1], r2 = rq[2];
                slot = jq;
                wide_init(T, S.scene, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r2.x, r2.y, r2.z));
                stk.reset();
            }
            q_lo += (uint32_t)n_idle < avail ? (uint32_t)n_idle : avail;
        }
        WDG_STAMP(dg_refill);
        // unwind, once per trip, straight-line: a lane whose group has no pending slot and that has no pending triangle takes
        // the newest stacked group, or has finished
        {
            const bool pop = !T.done & (T.tm == 0u) & ((T.gy >> 24) == 0u);
            const bool go = pop & (T.sp != 0);
            const bool fin = pop & (T.sp == 0);
            const int nsp = T.sp - 1;
            const bool refill = go & (nsp > 0);
            uint32_t n_x;
            float n_y, unused = 0.0f;
            stk.pop_masked(nsp - 1, refill, n_x, n_y, unused);
            T.gx = go ? T.top_x : T.gx;
            T.gy = go ? T.top_y : T.gy;
            T.sp = go ? nsp : T.sp;
            T.top_x = refill ? n_x : T.top_x;
            T.top_y = refill ? __float_as_uint(n_y) : T.top_y;
            if (fin) {
                *reinterpret_cast<float4 *>(L.hits + slot) = make_float4(__uint_as_float(T.best.k), T.best.b, T.best.c, T.best.k == RT_NONE ? 0.0f : T.best.t);
                T.done = true;
            }
        }
        WDG_STAMP(dg_pop);
        const bool waiting = !T.done && T.tm != 0u;
        const bool stepper = !T.done && T.tm == 0u && (T.gy >> 24) != 0u;
        const unsigned long long wm = __ballot(waiting), sm = __ballot(stepper);
        if ((wm | sm) == 0ull) {
            if (__ballot(!T.done) != 0ull)
                continue; // only lanes that just popped an exhausted group: unwind again
            if (exhausted)
                break;
            continue;
        }
        if (sm == 0ull || __popcll(wm) >= RT_WIDE_TRI_MIN) {
            wide_tri_batch<STATS>(T, blob, waiting, s_owner, s_min, s_bc, st);
#ifdef RT_DIAG_CYCLES
            dg_tri_n += 1, dg_tri_lanes += (unsigned long long)__popcll(wm);
#endif
            WDG_STAMP(dg_tri);
        } else {
            if (stepper)
                wide_node_step<STATS>(T, blob, stk, st);
#ifdef RT_DIAG_CYCLES
            dg_nodes_n += 1, dg_nodes_lanes += (unsigned long long)__popcll(sm);
#endif
            WDG_STAMP(dg_node);
        }
    }
#ifdef RT_DIAG_CYCLES
    if ((threadIdx.x & 63u) == 0u && L.diag) {
        unsigned long long *dg = reinterpret_cast<unsigned long long *>(L.diag);
        atomicAdd(dg + 0, dg_refill), atomicAdd(dg + 1, dg_pop), atomicAdd(dg + 2, dg_node), atomicAdd(dg + 3, dg_tri);
        atomicAdd(dg + 4, __builtin_amdgcn_s_memtime() - dg_start), atomicAdd(dg + 5, 1ull);
        atomicAdd(dg + 6, dg_nodes_n), atomicAdd(dg + 7, dg_nodes_lanes), atomicAdd(dg + 8, dg_tri_n), atomicAdd(dg + 9, dg_tri_lanes);
    }
#endif
    st.flush(L.stats);
}

// ------------------------------------------------------------------------------------------------ coherent packets
// Primary rays: a wave's 64 queue positions are 64 samples of one pixel (or of a few neighbours), rays that walk the same nodes.
// Here the wave walks the tree ONCE for all of them: the pending-group stack is wave-uniform (LDS, one column per wave), every
// node and triangle record is fetched once through the scalar cache, each lane tests the eight boxes / the triangle against
// its OWN ray and best hit, and the wave descends into a child when ANY lane hit it (front to back by the first lane's
// direction signs). A lane tests a triangle only if its own ray met that leaf slot's box, as in the per-lane kernel; its
// closest hit is therefore the same (it may see boxes the per-lane walk would have culled earlier, never fewer).
// No vector loads, no divergence between node and triangle work: it pays while the 64 rays stay together (the kernel counts
// node visits and the lanes that met the node; the host drops it for a configuration whose packets fall apart).
#ifndef RT_WIDE_PKT_CHUNK
#define RT_WIDE_PKT_CHUNK 256u
#endif
#ifndef RT_WIDE_PKT_WAVES_PER_SIMD
#define RT_WIDE_PKT_WAVES_PER_SIMD 6 /* 80 VGPRs: a lane carries its ray's precomputed slab constants (WRay, 12 registers); the kernel issues VALU ~80 % of the time, 6 waves feed that */
#endif
template <bool STATS> __global__ __launch_bounds__(256, RT_WIDE_PKT_WAVES_PER_SIMD) void wf_extend_wide_packet(const DevScene S, const WfLaunch L) {
    __shared__ uint2 s_stack_all[4][RT_MAX_STACK + 1]; // one pending group per level of the tree at most (depth <= RT_MAX_STACK)
    uint2 *s_stack = s_stack_all[threadIdx.x >> 6];
    LaneStats<STATS> st;
    const uint32_t n_in = L.counters[WF_CNT_IN];
    const uint32_t lane = threadIdx.x & 63u;
    const uint4 *blob = reinterpret_cast<const uint4 *>(S.scene.wide);
    unsigned long long n_trips = 0ull, n_lanes = 0ull; // wave-uniform
    for (;;) {
        uint32_t base = 0;
        if (lane == 0u)
            base = atomicAdd(L.counters + WF_CNT_TICKET, (uint32_t)RT_WIDE_PKT_CHUNK);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_in)
            break;
        for (uint32_t q0 = base; q0 < base + RT_WIDE_PKT_CHUNK && q0 < n_in; q0 += 64u) { // wave-uniform
            const uint32_t jq = q0 + lane;
            const bool have = jq < n_in;
            V3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f);
            WRay w = wide_ray(S.scene.grid, o, mk(1.f, 1.f, 1.f));
            uint32_t my_oct_inv = 7u;
            Hit best = Hit{RT_NONE, 0.f, 0.f, -RT_INF}; // a lane without a ray: an empty [EPS, -inf] range meets no box
            if (have) {
                const uint32_t j = L.order ? L.order[jq] & WF_ORDER_SLOT_MASK : jq;
                const float4 *rq = reinterpret_cast<const float4 *>(L.paths_in + j);
                const float4 r0 = rq[0], r1 = rq[1], r2 = rq[2];
                WTrav T;
                wide_init(T, S.scene, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r2.x, r2.y, r2.z));
                o = T.o, d = T.d, w = T.w, my_oct_inv = T.oct_inv;
                best.t = RT_INF;
            }
            const uint32_t oct_inv = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_oct_inv); // lane 0 of the packet always has a ray
            uint32_t gx = 0u, gy = S.scene.n_wide != 0u ? 0x80000000u : 0u; // wave-uniform
            int sp = 0;
            for (;;) {
                if ((gy >> 24) == 0u) {
                    if (sp == 0)
                        break;
                    --sp;
                    const uint2 g = s_stack[sp];
                    gx = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.x);
                    gy = (uint32_t)__builtin_amdgcn_readfirstlane((int)g.y);
                    continue;
                }
                const uint32_t bit = 31u - (uint32_t)__clz((int)gy);
                const uint32_t rest = gy ^ (1u << bit);
                const uint32_t slot = (bit - 24u) ^ oct_inv;
                uint32_t idx = gx + RT_WIDE_NODE_UNITS * (uint32_t)__popc(gy & 0xFFu & ((1u << slot) - 1u));
                if ((rest >> 24) != 0u && sp < RT_MAX_STACK) {
                    if (lane == 0u)
                        s_stack[sp] = make_uint2(gx, rest);
                    ++sp;
                }
                idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx); // the record's address is scalar: ONE 64-byte s_load for the whole wave
                typedef uint32_t U4v __attribute__((ext_vector_type(4)));
                typedef const __attribute__((address_space(4))) U4v *ConstU4;
                ConstU4 p = (ConstU4)(unsigned long long)(blob + idx);
                const U4v v0 = p[0], v2 = p[1], v3 = p[2], v4 = p[3];
                const uint4 n0 = make_uint4(v0.x, v0.y, v0.z, v0.w), n2 = make_uint4(v2.x, v2.y, v2.z, v2.w), n3 = make_uint4(v3.x, v3.y, v3.z, v3.w),
                            n4 = make_uint4(v4.x, v4.y, v4.z, v4.w);
                const WideHdr H = wide_decode(n0); // wave-uniform: scalar arithmetic
                const uint32_t imask = H.imask;
                const uint32_t h = wide_test8(H, n2, n3, n4, w, best.t);
                if (have) {
                    st.node();
                    st.box(8);
                }
                uint32_t Hany = 0u; // slots ANY lane hit
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    Hany |= __ballot((h >> i) & 1u) != 0ull ? (1u << i) : 0u;
                ++n_trips;
                n_lanes += (uint32_t)__popcll(__ballot(h != 0u));
                gx = H.base;
                gy = (wide_priority(Hany & imask, oct_inv) << 24) | imask;
                // triangles of the leaf slots some lane met, in record order; a lane tests those of the slots IT met
                uint32_t tm = wide_leaf_tris(Hany & ~imask, H.tri_mask);
                const uint32_t tall = H.tri_mask, tbase = H.base + RT_WIDE_NODE_UNITS * (uint32_t)__popc(imask);
                while (tm != 0u) {
                    const uint32_t b = (uint32_t)__ffs((int)tm) - 1u;
                    tm &= tm - 1u;
                    uint32_t k = tbase + RT_WIDE_TRI_UNITS * (uint32_t)__popc(tall & ((1u << b) - 1u));
                    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                    ConstF4 tp = as_const_f4(blob + k);
                    const F4v r0 = tp[0], r1 = tp[1], r2 = tp[2];
                    if ((h >> (b / 3u)) & 1u) {
                        st.tri();
                        V3 xs;
                        if (tri_hit(mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), o, d, EPS, xs) && best.t > xs.z) {
                            best.k = __float_as_uint(r2.w); // DevTri::pad of a blob record: its DevTri / DevAttr index
                            best.b = xs.x;
                            best.c = xs.y;
                            best.t = xs.z;
                        }
                    }
                }
            }
            if (have)
                *reinterpret_cast<float4 *>(L.hits + jq) = make_float4(__uint_as_float(best.k), best.b, best.c, best.k == RT_NONE ? 0.0f : best.t);
        }
    }
    if (lane == 0u && L.packet_census && n_trips != 0ull) {
        atomicAdd(L.packet_census, n_trips);
        atomicAdd(L.packet_census + 1, n_lanes);
    }
    st.flush(L.stats);
}

} // namespace

namespace rt {

hipError_t launch_extend_wide(const DevScene &S, const WfLaunch &L, bool packet, bool stats, int blocks, hipStream_t stream) {
    if (packet) {
        if (stats)
            return RT_LAUNCH_CHECKED((wf_extend_wide_packet<true>), dim3(blocks), dim3(256), 0, stream, S, L);
        return RT_LAUNCH_CHECKED((wf_extend_wide_packet<false>), dim3(blocks), dim3(256), 0, stream, S, L);
    }
    if (stats)
        return RT_LAUNCH_CHECKED((wf_extend_wide<true>), dim3(blocks), dim3(256), 0, stream, S, L);
    return RT_LAUNCH_CHECKED((wf_extend_wide<false>), dim3(blocks), dim3(256), 0, stream, S, L);
}

} // namespace rt

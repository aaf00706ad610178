// rt_wavefront.hip — the production render path (RT_RNG_DEVICE): a wavefront pipeline over PATHS.
//
// The reference renders pixel by pixel, sample by sample, bounce by bounce inside one recursive call chain
// (render_pixel -> trace_ray <-> shade, src/raytracer.h:555-627). On a 64-wide machine that nesting leaves most lanes
// idle: traversal lengths are heavy-tailed and a lane that finishes early waits for the wave's slowest ray before it may
// shade. Here a path = one (pixel, sample) and the recursion is cut into stages that each run over ALL live paths:
//
//   wf_generate : gen_ray (raytracer.h:527-538) for every path of the pass -> ray queue
//   per bounce (ray_depth times):
//     wf_extend : closest hit (BVH::intersect_ray, bvh.h:195-235) for every queued ray. Persistent wavefronts; a lane
//                 whose traversal ends stores its hit and is refilled from the queue (wave ballot + prefix count out of
//                 the wave's private chunk of queue positions, one ticket atomic per 128-ray chunk), so the wave stays
//                 dense whatever the spread of traversal lengths. Bounces >= 1 walk the queue in a coherence-sorted order.
//     wf_shade  : one shade() level (raytracer.h:555-591) per hit: texture fetches, sampling, pdfs (incl. the light-BVH
//                 traversal), BRDF. Finished paths fold their (emission, scale) frames back-to-front (the Horner order of
//                 raytracer.h:588-590) and store the sample; surviving paths are COMPACTED into the next ray queue with
//                 a wave ballot + prefix sum (one atomic per wave).
//   wf_resolve  : per pixel, the samples of the pass are added in sample order s = 0,1,2,... onto the running sum, so the
//                 float sum has exactly the reference's order (raytracer.h:621-626) although samples ran in parallel.
//
// Every path owns an xoshiro128++ stream seeded from (seed, pixel, sample) and consumes it in the reference's draw
// order; the stream's state, the remaining depth and the count of pending frames travel with the ray through the queues
// (WfPath), so the image is independent of queue order, tiling and GPU count, and bit-identical to the persistent
// megakernel of rt_kernels.hip and to the CPU oracle in device-RNG mode.
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "rt_device_lib.h"
#include "rt_kernels.h"

namespace {

#ifndef RT_EXT_WAVES_PER_SIMD
#define RT_EXT_WAVES_PER_SIMD 6 /* <= 80 VGPRs; measured best with RT_EXT_LDS_DEPTH 6 (tools_sweep.sh) */
#endif
#ifndef RT_EXT_LDS_DEPTH
#define RT_EXT_LDS_DEPTH 6 /* LDS part of the traversal stack in wf_extend: 6 -> 26 KB/block -> 6 blocks (24 waves) per CU */
#endif
#ifndef RT_EXT_GB_LDS_DEPTH
#define RT_EXT_GB_LDS_DEPTH 9 /* global-best traversal: 8-byte frames {ref, d_far} -> 9 positions in the LDS the reference traversal's 6 x 12 B take */
#endif
#ifndef RT_SHADE_WAVES_PER_SIMD
#define RT_SHADE_WAVES_PER_SIMD 4
#endif
#ifndef RT_SHADE_BLOCKS_PER_CU
#define RT_SHADE_BLOCKS_PER_CU 8 /* grid of the grid-stride kernels (wf_shade, wf_extend_prims) */
#endif
#ifndef RT_SHADE_LDS_DEPTH
#ifndef RT_SHADE_RECLASS
#define RT_SHADE_RECLASS 1 /* wf_shade hands its block's hits to the lanes sorted by sampler class */
#endif
#define RT_SHADE_LDS_DEPTH 4 /* only the light-BVH traversal of bvh_mix_dist::pdf uses a stack in wf_shade */
#endif
using ShadeStack = StackMemT<RT_SHADE_LDS_DEPTH>;
#ifndef RT_EXT_POP_ONCE
#define RT_EXT_POP_ONCE 1 /* bounded unwind: one stack pop per trip instead of an inner loop until no lane unwinds */
#endif
#ifndef RT_EXT_CHUNK
#define RT_EXT_CHUNK 64u /* queue positions a wave takes per ticket atomic (128 until the tickets were partitioned: with eight heads the atomics are cheap
                            and the last ticket of a part is one batch of work, not two: + 0.7 % S-sponza, + 1 % S-10M, profiles/r03_variants.txt item 21) */
#endif
#ifndef RT_EXT_REFILL_MIN
#define RT_EXT_REFILL_MIN 16 /* refill a wave's idle lanes once this many have finished (a refill stalls the wave on the ray loads) */
#endif

DEV uint32_t wf_global_pixel(const WfLaunch &L, uint32_t local_pixel) {
    const uint32_t local_block = local_pixel / L.shard_block;
    const uint32_t within = local_pixel - local_block * L.shard_block;
    return (local_block * L.shard_count + L.shard_index) * L.shard_block + within;
}

// ------------------------------------------------------------------------------------------------ generate
template <bool STATS> __global__ __launch_bounds__(256) void wf_generate(const DevScene S, const WfLaunch L) {
    LaneStats<STATS> st;
    const V3 cam_pos = ld3(S.cam_pos), cam_right = ld3(S.cam_right), cam_up = ld3(S.cam_up), cam_fwd = ld3(S.cam_fwd);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        L.counters[WF_CNT_IN] = L.n_paths; // the first bounce's queue is the identity: slot i holds path i
#ifdef RT_DIAG
    if (STATS && threadIdx.x == 0 && blockIdx.x == 0)
        g_diag = (DevStats *)L.diag; // set a launch ahead of the kernels that count
#endif
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < L.n_paths; i += gridDim.x * blockDim.x) {
        const uint32_t lp = i / L.pass_samples;
        const uint32_t ds = i - lp * L.pass_samples;
        const uint32_t pix = wf_global_pixel(L, L.first_pixel + lp);
        const uint32_t s = L.first_sample + ds;
        Rng<RT_RNG_DEVICE> rng;
        rt_xoshiro_seed(&rng.g, L.seed, pix, s);
        const uint32_t x = pix % L.width, y = pix / L.width;
        float ox = uniform_real(rng, 0.0f, 1.0f);
        float oy = uniform_real(rng, 0.0f, 1.0f);
        float sx = (2 * ((float)(int)x + ox) / (float)L.width - 1) * L.tan_x;
        float sy = (2 * ((float)(int)y + oy) / (float)L.height - 1) * L.tan_y;
        V3 rd = norm(sx * cam_right - sy * cam_up + 1.0f * cam_fwd);
        float4 *rq = reinterpret_cast<float4 *>(L.paths_in + i);
        rq[0] = make_float4(cam_pos.x, cam_pos.y, cam_pos.z, rd.x);
        rq[1] = make_float4(rd.y, rd.z, __uint_as_float(i | (next_shade_class(rng, S.lights.n_tris != 0) << WF_ORDER_CLASS_SHIFT)), __uint_as_float(L.ray_depth)); // path id (+ class); full budget, no pending frames
        rq[2] = make_float4(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z, __uint_as_float(ray_fast_ok_ray(cam_pos, rd) ? 1u : 0u));
        *reinterpret_cast<uint4 *>(rq + 3) = make_uint4(rng.g.s[0], rng.g.s[1], rng.g.s[2], rng.g.s[3]);
        st.cast(); // ray_depth >= 1: trace_ray casts (raytracer.h:600)
    }
    st.flush(L.stats);
}

// ------------------------------------------------------------------------------------------------ extend
// Closest hit for every queued ray. Two kinds of work alternate inside a wave instead of being interleaved:
//   * node steps   : lanes standing on an inner node test its two child boxes (trav_step_inner_fast; trav_step_core for a
//                    wave with a guarded ray or a big-leaf walker), lanes that reached a leaf wait;
//   * leaf batches : once enough lanes wait on leaves (or nobody is left on inner nodes) the wave tests ALL their
//                    triangles together: the (ray, triangle) pairs of the waiting lanes are laid out densely over the
//                    64 lanes (prefix sum of the leaf sizes), each lane fetches "its" ray from the owning lane with
//                    cross-lane reads and runs one triangle test; a leaf's result is the minimum of a 64-bit key
//                    (t bits, triangle index) reduced with LDS atomics, i.e. smallest t and, on equal t, the
//                    lowest triangle index — exactly the leaf loop's strict-less replacement order (bvh.h:200-204,132).
// This removes the inner-node / triangle divergence of a one-record-per-lane step (about 58 % / 42 % of the lanes) and
// packs the triangle tests: a wave does ~40 node tests or ~48 triangle tests per pass instead of ~27 + ~20.
template <bool STATS>
DEV void leaf_batch(Trav &T, const DevBvh &bvh, bool at_leaf, uint16_t *s_owner, unsigned long long *s_min, float2 *s_bc, LaneStats<STATS> &st) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = at_leaf ? RT_LEAF_CNT(T.cur) : 0u;
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) { // exclusive prefix sum of n over the wave, one ballot per bit plane
        const unsigned long long m = __ballot((n >> b) & 1u);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        off += below << b;
        total += (uint32_t)__popcll(m) << b;
    }
    // Owner table: position off + t belongs to (lane, t) for t < n. Every waiting lane writes ALL RT_LEAF_COOP_MAX
    // entries, highest t first, without a per-entry predicate: an entry with t >= n lands on position off' + t' of a
    // later lane (off' > off, hence t' < t), whose own store of that position is issued LATER (a wave's LDS
    // instructions execute in order) and wins; within one instruction the waiting lanes' positions are distinct
    // (their offsets increase strictly). Positions >= total are never read; the table has room for the overshoot.
    if (at_leaf) {
#pragma unroll
        for (int t = RT_LEAF_COOP_MAX - 1; t >= 0; --t) {
            s_owner[off + t] = (uint16_t)(lane | ((uint32_t)t << 8));
            asm volatile("" ::: "memory"); // keep the stores in this order (compiler and machine scheduler)
        }
        s_min[lane] = ~0ull;
    }
    __threadfence_block();
    const uint32_t k0 = T.cur & RT_LEAF_BEGIN_MASK;
    DIAG(13, 1);
    DIAG(14, (unsigned long long)__popcll(__ballot(at_leaf)));
    DIAG(16, total);
    for (uint32_t q0 = 0; q0 < total; q0 += 64u) { // wave-uniform trip count
        DIAG(15, 1);
        const uint32_t q = q0 + lane;
        const bool valid = q < total;
        const uint32_t ow = valid ? (uint32_t)s_owner[q] : 0u;
        const int src = (int)(ow & 63u);
        const uint32_t kk = (uint32_t)__shfl((int)k0, src) + (ow >> 8);
        const V3 o = mk(__shfl(T.o.x, src), __shfl(T.o.y, src), __shfl(T.o.z, src));
        const V3 d = mk(__shfl(T.d.x, src), __shfl(T.d.y, src), __shfl(T.d.z, src));
        if (valid) {
            const float4 *p = reinterpret_cast<const float4 *>(bvh.tris + kk);
            const float4 r0 = p[0], r1 = p[1], r2 = p[2];
            st.tri();
            V3 xs;
            if (tri_hit(mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), o, d, EPS, xs)) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(xs.z) << 32) | (unsigned long long)kk;
                DIAG(17, 1);
                atomicMin(&s_min[src], key);
                __threadfence_block();
                if (s_min[src] == key) // this pair leads its leaf so far: publish its barycentrics
                    s_bc[src] = make_float2(xs.x, xs.y);
            }
        }
    }
    __threadfence_block();
    if (at_leaf) {
        st.node(); // one BVH::intersect_ray invocation on the leaf node
        const unsigned long long key = s_min[lane];
        if (key != ~0ull) {
            const float t = __uint_as_float((uint32_t)(key >> 32));
            const float2 bc = s_bc[lane];
            if (T.best.k == RT_NONE || T.best.t > t) {
                T.best.k = (uint32_t)key;
                T.best.b = bc.x;
                T.best.c = bc.y;
                T.best.t = t;
            }
            T.t_loc = fminf(T.t_loc, t);
        }
        T.cur = T_POP; // unwound by the caller's trav_pop_wave
    }
}

#ifndef RT_EXT_LEAF_MIN
#define RT_EXT_LEAF_MIN 20 /* run a leaf batch once this many lanes wait on a leaf */
#endif

template <bool STATS, bool GB> __global__ __launch_bounds__(256, RT_EXT_WAVES_PER_SIMD) void wf_extend(const DevScene S, const WfLaunch L) {
    constexpr int DEPTH = GB ? RT_EXT_GB_LDS_DEPTH : RT_EXT_LDS_DEPTH, WORDS = GB ? 2 : 3;
    __shared__ uint32_t s_stack[STACK_LDS_DWORDS_W(DEPTH, WORDS)];
    __shared__ uint16_t s_owner_all[4][64 * RT_LEAF_COOP_MAX + RT_LEAF_COOP_MAX]; // + overshoot of the unpredicated owner stores
    __shared__ unsigned long long s_min_all[4][64];
    __shared__ float2 s_bc_all[4][64];
    const uint32_t wave = threadIdx.x >> 6;
    uint16_t *s_owner = s_owner_all[wave];
    unsigned long long *s_min = s_min_all[wave];
    float2 *s_bc = s_bc_all[wave];
    LaneStats<STATS> st;
    RT_DECLARE_RING_STACK_W(stk, DEPTH, WORDS, s_stack, L.stack_overflow, L.stack_stride);
#ifdef RT_DIAG
    if (STATS && threadIdx.x == 0 && blockIdx.x == 0)
        g_diag = (DevStats *)L.diag;
#endif
    const uint32_t n_in = L.counters[WF_CNT_IN];
    Trav T;
    T.cur = T_DONE;
    T.sp = 0;
    uint32_t slot = RT_NONE;
    bool exhausted = n_in == 0; // wave-uniform
    uint32_t q_lo = 0, q_hi = 0; // this wave's private range of queue positions
    TicketState tks = ticket_init();
#ifdef RT_DIAG_CYCLES
    // section census of the persistent loop (development build only, no other instrumentation): wave cycles between s_memtime stamps
    unsigned long long dg_refill = 0, dg_node = 0, dg_leaf = 0, dg_pop = 0, dg_store = 0, dg_t = __builtin_amdgcn_s_memtime();
    const unsigned long long dg_start = dg_t;
#define DG_STAMP(acc)                                             \
    do {                                                          \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        acc += now_ - dg_t;                                       \
        dg_t = now_;                                              \
    } while (0)
#else
#define DG_STAMP(acc) do { } while (0)
#endif
    for (;;) {
        const bool idle = T.cur == T_DONE;
        const unsigned long long im = __ballot(idle);
        const int n_idle = __popcll(im);
        if (!exhausted && (n_idle >= RT_EXT_REFILL_MIN || n_idle == (int)__popcll(__ballot(1)))) {
            // refill from the wave's private ticket range [q_lo, q_hi); a new range of RT_EXT_CHUNK queue positions is
            // taken with ONE atomic when it runs dry (a single ticket word saturates near 90 M atomics/s, so tickets
            // are taken per chunk, not per refill)
            if (q_lo == q_hi)
                exhausted = !ticket_take(L.counters, n_in, (uint32_t)RT_EXT_CHUNK, tks, q_lo, q_hi); // false: the queue is used up
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
            const uint32_t avail = q_hi - q_lo;
            if (idle && rank < avail) {
                const uint32_t jq = q_lo + rank;
                const uint32_t j = L.order ? L.order[jq] & WF_ORDER_SLOT_MASK : jq; // coherence-sorted processing order
                const float4 *rq = reinterpret_cast<const float4 *>(L.paths_in + j);
                const float4 r0 = rq[0], r1 = rq[1], r2 = rq[2];
                slot = jq; // the hit goes to the queue POSITION (see WfLaunch::hits)
                trav_init_stored<GB>(T, S.scene, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r2.x, r2.y, r2.z), __float_as_uint(r2.w) != 0u);
                stk.reset();
                if (T.cur == T_DONE) // no geometry at all: immediate miss
                    *reinterpret_cast<float4 *>(L.hits + jq) = make_float4(__uint_as_float(RT_NONE), 0.f, 0.f, 0.f);
            }
            q_lo += (uint32_t)n_idle < avail ? (uint32_t)n_idle : avail;
        }
        DIAG(12, 1);
        DG_STAMP(dg_refill);
#if RT_EXT_POP_ONCE
        // Bounded unwind: ONE stack pop per trip for every lane that has to unwind (a lane whose pop ends in a pruned far
        // child pops again next trip and sits out one node step: ~1 in 5 unwinding lanes). The unwind used to be a loop that
        // ran until no lane of the wave was left in T_POP: 1.2 iterations per trip at ~7 of 64 lanes, each a full LDS round
        // trip, plus the loop's own header and exit code — 22 % of the kernel's wave cycles (profiles/r02_extend_sections.txt).
        // (Tried and not kept: issuing the pop's LDS reads here and consuming them only behind the node fetch — 3 more live
        // VGPRs and the extra predicate traffic cost more than the hidden LDS round trip: 233.7 vs 238.5 Msamples/s.)
        const bool was_live = T.cur != T_DONE;
        trav_pop_once<GB>(T, stk);
        if (was_live && T.cur == T_DONE)
            *reinterpret_cast<float4 *>(L.hits + slot) = make_float4(__uint_as_float(T.best.k), T.best.b, T.best.c, T.best.t);
        DG_STAMP(dg_pop);
        const bool active = T.cur != T_DONE;
        const bool popping = T.cur == T_POP; // note: T_POP has the leaf bit set, it must be told apart first
        const bool at_leaf = active && !popping && (T.cur & RT_LEAF_FLAG) != 0 && RT_LEAF_CNT(T.cur) != 0;
        const bool stepper = active && !popping && !at_leaf; // inner node, or a big leaf walked triangle by triangle
        const unsigned long long lm = __ballot(at_leaf), sm = __ballot(stepper);
        if ((lm | sm) == 0ull) {
            if (__ballot(popping) != 0ull)
                continue; // only unwinding lanes left: pop again
            if (exhausted)
                break;
            continue;
        }
#else
        const bool active = T.cur != T_DONE;
        const bool at_leaf = active && (T.cur & RT_LEAF_FLAG) != 0 && RT_LEAF_CNT(T.cur) != 0;
        const bool stepper = active && !at_leaf; // inner node, or a big leaf walked triangle by triangle
        const unsigned long long lm = __ballot(at_leaf), sm = __ballot(stepper);
        if ((lm | sm) == 0ull) {
            if (exhausted)
                break;
            continue;
        }
#endif
        if (sm == 0ull || __popcll(lm) >= RT_EXT_LEAF_MIN) {
            leaf_batch<STATS>(T, S.scene, at_leaf, s_owner, s_min, s_bc, st);
            DG_STAMP(dg_leaf);
        } else {
            // the common wave: every stepping lane is on an inner node with the fast-division guarantees -> straight-line
            // node step; a wave with a big-leaf walker or a guarded ray takes the general step
            const bool plain = (T.cur & RT_LEAF_FLAG) == 0 && T.fast;
            if (__ballot(stepper && !plain) == 0ull) {
                if (stepper) {
                    DIAG(18, 1);
                    DIAG_LANES(19);
                    trav_step_inner_fast<STATS, GB>(T, S.scene, stk, EPS, st);
                }
            } else if (stepper) {
                trav_step_core<STATS, GB>(T, S.scene, stk, EPS, st);
            }
            DG_STAMP(dg_node);
        }
#if !RT_EXT_POP_ONCE
        trav_pop_wave<GB>(T, stk); // unwind after a leaf batch or a node step, all lanes of the wave together
        DG_STAMP(dg_pop);
        if (active && T.cur == T_DONE)
            *reinterpret_cast<float4 *>(L.hits + slot) = make_float4(__uint_as_float(T.best.k), T.best.b, T.best.c, T.best.t);
        DG_STAMP(dg_store);
#endif
    }
#ifdef RT_DIAG_CYCLES
    if ((threadIdx.x & 63u) == 0u && L.diag) {
        unsigned long long *dg = reinterpret_cast<unsigned long long *>(L.diag);
        atomicAdd(dg + 20, dg_refill);
        atomicAdd(dg + 21, dg_node);
        atomicAdd(dg + 22, dg_leaf);
        atomicAdd(dg + 23, dg_pop);
        atomicAdd(dg + 24, dg_store);
        atomicAdd(dg + 25, __builtin_amdgcn_s_memtime() - dg_start);
        atomicAdd(dg + 26, 1ull);
    }
#endif
    st.flush(L.stats);
}

// ------------------------------------------------------------------------------------------------ extend: coherent packets
// Primary rays. A wave's 64 queue positions are 64 samples of one pixel (or a few neighbouring pixels): rays that visit almost
// the same nodes. wf_extend lets its lanes drift apart (each lane refills on its own), so it pays one vector-L1 access per lane
// and 16-byte piece for them like for any other ray, and the L1 access rate is its roof (profiles/r02_l1_roof.txt). Here the
// wave stays a packet: all 64 rays start together, and every trip serves ONE record — the smallest pending node or leaf
// reference over the lanes (inner nodes before leaves, earlier nodes first, so stragglers catch up and lanes re-join) —
// fetched once through the scalar cache and applied by the lanes standing on it. Each lane still runs its own traversal
// state machine (trav_inner_apply / the leaf loop of trav_step_core / trav_pop_once, its own stack, its own near/far order
// and pruning), only WHEN a lane advances is decided per wave: results, order of strict-less replacements and counters are
// the per-lane kernel's by construction. No vector loads on the traversal path at all.
#ifndef RT_PKT_CHUNK
#define RT_PKT_CHUNK 256u /* queue positions per ticket atomic (4 packets) */
#endif
template <bool STATS, bool GB> __global__ __launch_bounds__(256, RT_EXT_WAVES_PER_SIMD) void wf_extend_packet(const DevScene S, const WfLaunch L) {
    constexpr int DEPTH = GB ? RT_EXT_GB_LDS_DEPTH : RT_EXT_LDS_DEPTH, WORDS = GB ? 2 : 3;
    __shared__ uint32_t s_stack[STACK_LDS_DWORDS_W(DEPTH, WORDS)];
    LaneStats<STATS> st;
    RT_DECLARE_RING_STACK_W(stk, DEPTH, WORDS, s_stack, L.stack_overflow, L.stack_stride);
    const uint32_t n_in = L.counters[WF_CNT_IN];
    const uint32_t lane = threadIdx.x & 63u;
#ifdef RT_DIAG
    if (STATS && threadIdx.x == 0 && blockIdx.x == 0)
        g_diag = (DevStats *)L.diag;
#endif
    Trav T;
    T.o = T.d = T.r = mk(0.f, 0.f, 0.f);
    T.cur = T_DONE;
    T.sp = 0;
    T.t_loc = GB ? RT_INF : RT_NAN;
    T.best = Hit{RT_NONE, 0.f, 0.f, 0.f};
    T.fast = false;
    T.top_ref = 0u;
    T.top_d = T.top_loc = 0.f;
    unsigned long long n_trips = 0ull, n_lanes = 0ull; // wave-uniform
    for (;;) {
        uint32_t base = 0;
        if (lane == 0u)
            base = atomicAdd(L.counters + WF_CNT_TICKET, (uint32_t)RT_PKT_CHUNK);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_in)
            break;
        for (uint32_t q0 = base; q0 < base + RT_PKT_CHUNK && q0 < n_in; q0 += 64u) { // wave-uniform
            const uint32_t jq = q0 + lane;
            const bool have = jq < n_in;
            T.cur = T_DONE;
            if (have) {
                const uint32_t j = L.order ? L.order[jq] & WF_ORDER_SLOT_MASK : jq;
                const float4 *rq = reinterpret_cast<const float4 *>(L.paths_in + j);
                const float4 r0 = rq[0], r1 = rq[1], r2 = rq[2];
                trav_init_stored<GB>(T, S.scene, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r2.x, r2.y, r2.z), __float_as_uint(r2.w) != 0u);
                stk.reset();
            }
            for (;;) {
                trav_pop_wave<GB>(T, stk); // every lane that has to unwind does, until none is left in T_POP
                const uint32_t target = wave_min_u32(T.cur); // T_DONE is the largest value a lane can hold here
                if (target == T_DONE)
                    break;
                const bool mine = T.cur == target;
                const unsigned long long mm = __ballot(mine);
                if (mm == 0ull) // cannot happen (the minimum is some lane's value); never spin on a wrong reduction
                    break;
                ++n_trips; // how coherent the packets are: the host keeps or drops this kernel on lanes per trip
                n_lanes += (uint32_t)__popcll(mm);
                DIAG(9, 1); // development census (tools/diag_packet.py): trips, lanes served, leaf trips
                DIAG(10, (unsigned long long)__popcll(mm));
                DIAG(11, (target & RT_LEAF_FLAG) ? 1ull : 0ull);
                if ((target & RT_LEAF_FLAG) == 0u) {
                    // the record's address must stay a scalar: inside `if (mine)` the compiler knows T.cur == target and would
                    // otherwise address the node through the lane's own T.cur (a vector load per lane)
                    uint32_t node_index = target;
                    asm volatile("" : "+s"(node_index));
                    ConstF4 p = as_const_f4(S.scene.nodes + node_index);
                    const F4v r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
                    if (mine)
                        trav_inner_apply<STATS, GB>(T, stk, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), mk(r2.y, r2.z, r2.w), __float_as_uint(r3.x),
                                                __float_as_uint(r3.y), EPS, st);
                } else { // a leaf: its triangles in index order, strict-less replacement (bvh.h:200-204,132)
                    const uint32_t cnt = RT_LEAF_CNT(target);
                    uint32_t k = target & RT_LEAF_BEGIN_MASK;
                    for (uint32_t i = 0;; ++i, ++k) {
                        ConstF4 p = as_const_f4(S.scene.tris + k);
                        const F4v r0 = p[0], r1 = p[1], r2 = p[2];
                        const uint32_t flags = __float_as_uint(r2.z);
                        if (mine) {
                            if (flags & 2u)
                                st.node(); // one BVH::intersect_ray invocation on the leaf node
                            st.tri();
                            V3 xs;
                            if (tri_hit(mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), T.o, T.d, EPS, xs)) {
                                if (T.best.k == RT_NONE || T.best.t > xs.z) {
                                    T.best.k = k;
                                    T.best.b = xs.x;
                                    T.best.c = xs.y;
                                    T.best.t = xs.z;
                                }
                                T.t_loc = fminf(T.t_loc, xs.z);
                            }
                        }
                        if (cnt != 0u ? i + 1u == cnt : (flags & 1u) != 0u)
                            break;
                    }
                    if (mine)
                        T.cur = T_POP;
                }
            }
            if (have)
                *reinterpret_cast<float4 *>(L.hits + jq) = make_float4(__uint_as_float(T.best.k), T.best.b, T.best.c, T.best.t);
        }
    }
    if (lane == 0u && L.packet_census && n_trips != 0ull) {
        atomicAdd(L.packet_census, n_trips);
        atomicAdd(L.packet_census + 1, n_lanes);
    }
    st.flush(L.stats);
}

// ------------------------------------------------------------------------------------------------ extend: analytic primitives
// Scenes of the scene-txt front end may hold analytic primitives (ELLIPSOID / PLANE, include/rt_primspec.h) next to their
// triangles. They have no BVH (a plane is unbounded; BASELINE config 2 is "intersect kernel only, no BVH"): every queued
// ray tests all of them in index order and keeps the nearer of (BVH hit, primitive hit), strict-less as bvh.h:132.
// One lane per ray, records coalesced; launched only when the scene has such primitives.
__global__ __launch_bounds__(256) void wf_extend_prims(const DevScene S, const WfLaunch L) {
    const uint32_t n_in = L.counters[WF_CNT_IN];
    for (uint32_t jq = blockIdx.x * blockDim.x + threadIdx.x; jq < n_in; jq += gridDim.x * blockDim.x) {
        const uint32_t j = L.order ? L.order[jq] & WF_ORDER_SLOT_MASK : jq;
        const float4 *rq = reinterpret_cast<const float4 *>(L.paths_in + j);
        const float4 r0 = rq[0], r1 = rq[1];
        const float4 hq = *reinterpret_cast<const float4 *>(L.hits + jq);
        Hit h;
        h.k = __float_as_uint(hq.x), h.b = hq.y, h.c = hq.z, h.t = hq.w;
        const uint32_t k0 = h.k;
        prims_closest(S, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), h);
        if (h.k != k0)
            *reinterpret_cast<float4 *>(L.hits + jq) = make_float4(__uint_as_float(h.k), h.b, h.c, h.t);
    }
}

// ------------------------------------------------------------------------------------------------ shade
// LIGHTS_LDS: the light BVH (inner nodes, triangles, aux) is small enough (DevBvh::lds_inner) to be staged in LDS once per
// block; bvh_mix_dist's sample and pdf then read it there: a dozen dependent L1 round trips per hit become LDS reads.
template <bool STATS, bool LIGHTS_LDS, bool ENV> __global__ __launch_bounds__(256, RT_SHADE_WAVES_PER_SIMD) void wf_shade(const DevScene S, const WfLaunch L) {
    __shared__ float s_lin[256];
    __shared__ float s_gam[256];
    __shared__ uint32_t s_stack[STACK_LDS_DWORDS_FOR(RT_SHADE_LDS_DEPTH)];
    __shared__ float4 s_lights[LIGHTS_LDS ? RT_SHADE_LIGHTS_F4 : 1];
#if RT_SHADE_RECLASS
    __shared__ uint2 s_perm[4][256]; // per wave: (queue position, queue slot) of its 256 positions, sorted by sampler class
#endif
    LightTabs LT = light_tabs_global(S);
    if (LIGHTS_LDS) {
        const uint32_t n_node_f4 = 4u * (S.lights.lds_inner - 1u), n_tri_f4 = 3u * S.lights.n_tris;
        for (uint32_t i = threadIdx.x; i < n_node_f4; i += blockDim.x)
            s_lights[i] = LT.nodes[i];
        for (uint32_t i = threadIdx.x; i < n_tri_f4; i += blockDim.x)
            s_lights[n_node_f4 + i] = LT.tris[i];
        for (uint32_t i = threadIdx.x; i < S.lights.n_tris; i += blockDim.x)
            s_lights[n_node_f4 + n_tri_f4 + i] = LT.aux[i];
        LT = LightTabs{s_lights, s_lights + n_node_f4, s_lights + n_node_f4 + n_tri_f4};
    }
    s_lin[threadIdx.x] = S.lut_linear[threadIdx.x];
    s_gam[threadIdx.x] = S.lut_gamma[threadIdx.x];
#ifdef RT_DIAG_SHADE
    if (threadIdx.x < 4u * SD_N)
        (&g_sd_cyc[0][0])[threadIdx.x] = 0ull, (&g_sd_lanes[0][0])[threadIdx.x] = 0ull;
    if ((threadIdx.x & 63u) == 0u)
        g_sd_t[threadIdx.x >> 6] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
    LaneStats<STATS> st;
    RT_DECLARE_STACK(stk, RT_SHADE_LDS_DEPTH, s_stack);
    const bool has_lights = S.lights.n_tris != 0; // raytracer.h:449-453
    const uint32_t n_in = L.counters[WF_CNT_IN];
    const uint32_t n_slots = (n_in + 63u) >> 6; // wave slots of this launch: positions 64w .. 64w+63
    // one shade() level for the hit at queue position jq (the ray in queue slot j), as the lanes of one wave; `wave_slot` is the wave slot
    // (64 positions) this trip stands for: survivors go to the sub-queue of that slot (rt_device_types.h, WF_STRIPES)
    auto shade_wave = [&](const uint32_t jq, const uint32_t j, const uint32_t wave_slot) {
        const bool active = jq < n_in;
        bool survive = false;
        float4 nr0 = make_float4(0.f, 0.f, 0.f, 0.f), nr1 = nr0, nr2 = nr0;
        uint4 nrng = make_uint4(0u, 0u, 0u, 0u);
        if (active) {
            const float4 *rq = reinterpret_cast<const float4 *>(L.paths_in + j);
            const float4 r0 = rq[0], r1 = rq[1];
            const uint4 p0 = *reinterpret_cast<const uint4 *>(rq + 3);
            const float4 hq = *reinterpret_cast<const float4 *>(L.hits + jq);
            const uint32_t path = __float_as_uint(r1.z) & WF_ORDER_SLOT_MASK; // the top bits: this shade()'s sampler class (lane assignment above)
            Rng<RT_RNG_DEVICE> rng;
            rng.g.s[0] = p0.x, rng.g.s[1] = p0.y, rng.g.s[2] = p0.z, rng.g.s[3] = p0.w;
            uint32_t depth_left = __float_as_uint(r1.w) & 0xFFFFu, nb = __float_as_uint(r1.w) >> 16;
            Hit h;
            h.k = __float_as_uint(hq.x), h.b = hq.y, h.c = hq.z, h.t = hq.w;
            if (h.k != RT_NONE)
                depth_left -= 1; // shade(..., max_depth - 1)
            SD_STAMP(SD_LOAD);
            const ShadeResult sr = shade_hit<Rng<RT_RNG_DEVICE>, STATS, ENV>(S, LT, h, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), rng, has_lights, stk, s_lin, s_gam, st);
            SD_STAMP(SD_BRDF);
            bool terminal = sr.terminal;
            V3 term = sr.term;
            if (sr.push) { // emission + trace_ray(...) * scl (raytracer.h:588-590), folded when the path ends
                float4 *fw = reinterpret_cast<float4 *>(L.fold + ((size_t)nb * L.n_paths + path));
                fw[0] = make_float4(sr.emission.x, sr.emission.y, sr.emission.z, 0.f);
                fw[1] = make_float4(sr.scl.x, sr.scl.y, sr.scl.z, 0.f);
                ++nb;
            }
            if (!terminal && depth_left == 0) { // trace_ray(..., 0) returns (0,0,0) without casting (:596-598)
                terminal = true;
                term = mk(0, 0, 0);
            }
            if (terminal) {
                // The pending shade() frames are folded by wf_fold after the last bounce, not here: at any bounce only about a
                // quarter of a wave's paths end, each with its own number of frames, so the unwind ran at 16 of 64 lanes behind
                // dependent loads (14 % of this kernel's cycles). Leave the innermost value and the frame count.
                L.sample_out[path] = RtF4{term.x, term.y, term.z, __uint_as_float(nb)};
                st.sample();
                SD_STAMP(SD_FOLD);
            } else {
                survive = true;
                st.cast();
                nr0 = make_float4(sr.nro.x, sr.nro.y, sr.nro.z, sr.nrd.x);
                nr1 = make_float4(sr.nrd.y, sr.nrd.z, __uint_as_float(path | (next_shade_class(rng, has_lights) << WF_ORDER_CLASS_SHIFT)), __uint_as_float(depth_left | (nb << 16)));
                nr2 = make_float4(1.0f / sr.nrd.x, 1.0f / sr.nrd.y, 1.0f / sr.nrd.z, __uint_as_float(ray_fast_ok_ray(sr.nro, sr.nrd) ? 1u : 0u));
                nrng = make_uint4(rng.g.s[0], rng.g.s[1], rng.g.s[2], rng.g.s[3]);
            }
        }
        // compact the survivors of this wave into the next queue: ballot + prefix sum, one atomic per wave, on the counter of
        // the sub-queue this wave slot belongs to (rt_device_types.h, WF_STRIPES)
        const unsigned long long m = __ballot(survive);
        if (m != 0ull) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            uint32_t obase = 0;
            const int leader = __ffsll((long long)m) - 1;
            const uint32_t stripe = /* the wave's own slot: a sub-queue holds what ITS slots can emit */ (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_slot) % WF_STRIPES;
            if ((int)(threadIdx.x & 63u) == leader)
                obase = wf_stripe_base(stripe, n_slots) + atomicAdd(L.stripes + stripe * WF_STRIPE_WORDS, (uint32_t)__popcll(m));
            obase = __shfl(obase, leader);
            if (survive) {
                float4 *rw = reinterpret_cast<float4 *>(L.paths_out + obase + rank);
                rw[0] = nr0;
                rw[1] = nr1;
                rw[2] = nr2;
                *reinterpret_cast<uint4 *>(rw + 3) = nrng;
            }
        }
        SD_STAMP(SD_STORE);
    };
#if RT_SHADE_RECLASS
    // Which of shade()'s three samplers a hit runs is decided by its path's next draws alone (alpha coin, technique coin, mix pick:
    // raytracer.h:559,565,386): whoever wrote the path record left that class in the top bits of its path word, and the ray-order pass carried it along
    // in the top bits of `order`. A WAVE takes 256 queue positions at a time and hands them to its lanes sorted by class (a counting sort
    // over indices through the wave's own LDS window, before anything else is loaded, no block barrier): of its four trips one or two run
    // a single sampler and the others two instead of all three, and the rays aimed at a light walk the light BVH of bvh_mix_dist::pdf
    // side by side. Every wave still gets every class, so the waves of a block stay balanced. Which lane shades a hit changes no result.
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    // A queue too short to give every wave of the grid such a window is shaded 64 positions at a time (more waves, no reordering), and so
    // is one no sort ran over (primary rays; RT_SORT_OFF; the wide tree of a cache-resident scene): fetching the class from the records
    // ahead of the shading costs that kernel 6 % (measured, profiles/r04_variants.txt item 11); in the sort's payload it is free.
    const uint32_t win = (L.order_classed && n_in >= gridDim.x * 1024u) ? 256u : 64u;
    for (uint32_t wbase = (blockIdx.x * 4u + wv) * win; wbase < n_in; wbase += gridDim.x * 4u * win) { // wave-uniform
        uint32_t slot_of[4], cls[4], before[3] = {0u, 0u, 0u}, within[4];
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            const uint32_t jq = wbase + 64u * q + lane;
            slot_of[q] = jq, cls[q] = 3u; // class 3: the positions behind the queue's (or the window's) end
            if (64u * q < win && jq < n_in) {
                const uint32_t v = L.order ? L.order[jq] : jq;
                slot_of[q] = v & WF_ORDER_SLOT_MASK, cls[q] = v >> WF_ORDER_CLASS_SHIFT;
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) { // position inside the class: the class's members in earlier quarters, then in lower lanes
            const unsigned long long b0 = __ballot(cls[q] == 0u), b1 = __ballot(cls[q] == 1u), b2 = __ballot(cls[q] == 2u);
            const unsigned long long mine = cls[q] == 0u ? b0 : cls[q] == 1u ? b1 : cls[q] == 2u ? b2 : ~(b0 | b1 | b2);
            const uint32_t seen = cls[q] == 0u ? before[0] : cls[q] == 1u ? before[1] : cls[q] == 2u ? before[2] : 64u * q - before[0] - before[1] - before[2];
            within[q] = seen + __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
            before[0] += (uint32_t)__popcll(b0), before[1] += (uint32_t)__popcll(b1), before[2] += (uint32_t)__popcll(b2);
        }
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            const uint32_t start = cls[q] == 0u ? 0u : cls[q] == 1u ? before[0] : cls[q] == 2u ? before[0] + before[1] : before[0] + before[1] + before[2];
            s_perm[wv][start + within[q]] = make_uint2(wbase + 64u * q + lane, slot_of[q]);
        }
        __builtin_amdgcn_wave_barrier(); // the wave's LDS operations complete in order: its reads below see the writes above
        const uint32_t n_here = n_in - wbase < win ? n_in - wbase : win;
        for (uint32_t sub = 0; 64u * sub < n_here; ++sub) {
            const uint2 pj = s_perm[wv][64u * sub + lane];
            shade_wave(pj.x, pj.y, (wbase >> 6) + sub);
        }
        __builtin_amdgcn_wave_barrier();
    }
#else
    // wave-uniform trip count: the ballots in shade_wave must see the whole wave
    for (uint32_t base = blockIdx.x * blockDim.x; base < n_in; base += gridDim.x * blockDim.x) {
        const uint32_t jq = base + threadIdx.x; // queue position: where wf_extend left this ray's hit
        shade_wave(jq, jq < n_in ? (L.order ? L.order[jq] & WF_ORDER_SLOT_MASK : jq) : 0u, jq >> 6);
    }
#endif
#ifdef RT_DIAG_SHADE
    __syncthreads();
    if (threadIdx.x < 4u * SD_N && L.diag) { // census words 0..11: wave cycles per section, 12..23: lane-weighted cycles (summed over waves)
        unsigned long long *dg = reinterpret_cast<unsigned long long *>(L.diag);
        atomicAdd(dg + (threadIdx.x % SD_N), (&g_sd_cyc[0][0])[threadIdx.x]);
        atomicAdd(dg + SD_N + (threadIdx.x % SD_N), (&g_sd_lanes[0][0])[threadIdx.x]);
    }
#endif
    st.flush(L.stats);
}

// Ray-ordering key for secondary bounces: Morton code of the origin's cell in a 64^3 grid over the scene bounds (18 bits)
// followed by the direction octant (3 bits). Rays that start close together and head the same way end up in the same
// wave of wf_extend, so their gathers touch the same nodes (cache lines, L2 residency). Ordering never changes a result:
// every path's arithmetic is independent of where it sits in the queue.
DEV uint32_t spread3(uint32_t v) { // 6 bits -> every third bit
    v &= 63u;
    v = (v | (v << 8)) & 0x0300Fu;
    v = (v | (v << 4)) & 0x030C3u;
    v = (v | (v << 2)) & 0x09249u;
    return v;
}
// `direct`: no sort follows (sorting off or a tiny queue): the identity order over the dense index goes straight to sort_vals[1].
// Either way this pass turns the dense ray index j < n into the physical slot of paths_in (sub-queue regions, WF_STRIPES).
// `bound` >= n is what the HOST knows about the queue size when it launches this bounce (the size of the previous bounce's
// queue, read back without stalling the device): the sort that follows runs over `bound` pairs, so positions [n, bound) get
// the largest key and end up behind every real ray (nobody reads their order entries: consumers stop at n).
__global__ __launch_bounds__(256) void wf_sort_keys(const DevScene S, const WfLaunch L, int direct, uint32_t bound) {
    __shared__ uint32_t s_run[WF_STRIPES + 1u];
    const uint32_t n = L.counters[WF_CNT_IN], n_slots = L.counters[WF_CNT_SLOTS];
    if (threadIdx.x <= WF_STRIPES)
        s_run[threadIdx.x] = L.stripes[WF_STRIPES * WF_STRIPE_WORDS + threadIdx.x];
    __syncthreads();
    if (!direct)
        for (uint32_t j = n + blockIdx.x * blockDim.x + threadIdx.x; j < bound; j += gridDim.x * blockDim.x) {
            L.sort_keys[0][j] = 0xFFFFFFFFu;
            L.sort_vals[0][j] = 0u;
        }
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        uint32_t pos = j;
        if (n_slots != 0u) { // run k holds dense indices [s_run[k], s_run[k+1])
            uint32_t k = 0;
#pragma unroll
            for (uint32_t step = WF_STRIPES / 2u; step != 0u; step >>= 1)
                if (s_run[k + step] <= j)
                    k += step;
            pos = wf_stripe_base(k, n_slots) + (j - s_run[k]);
        }
        const float4 *rq = reinterpret_cast<const float4 *>(L.paths_in + pos);
        if (direct) { // no record is read here: WfLaunch::order_classed = 0
            L.sort_vals[1][j] = pos;
            continue;
        }
        const float4 r0 = rq[0], r1 = rq[1];
        const uint32_t val = pos | (__float_as_uint(r1.z) & ~WF_ORDER_SLOT_MASK); // the class bits of the path word ride along above the slot (WfLaunch::order)
        const float fx = (r0.x - S.bounds_lo[0]) * S.bounds_inv[0], fy = (r0.y - S.bounds_lo[1]) * S.bounds_inv[1], fz = (r0.z - S.bounds_lo[2]) * S.bounds_inv[2];
        const uint32_t cx = (uint32_t)fminf(fmaxf(fx * 64.0f, 0.0f), 63.0f), cy = (uint32_t)fminf(fmaxf(fy * 64.0f, 0.0f), 63.0f), cz = (uint32_t)fminf(fmaxf(fz * 64.0f, 0.0f), 63.0f);
        const uint32_t morton = spread3(cx) | (spread3(cy) << 1) | (spread3(cz) << 2);
        const uint32_t oct = (r0.w < 0.0f ? 1u : 0u) | (r1.x < 0.0f ? 2u : 0u) | (r1.y < 0.0f ? 4u : 0u);
        uint32_t key = (morton << 3) | oct; // mode 1: 64^3 cell, then octant
        if (L.sort_mode == 2) { // 16^3 cell (12 bits), then a 9-bit direction code (3 bits per component)
            const uint32_t dxq = (uint32_t)fminf(fmaxf((r0.w * 0.5f + 0.5f) * 8.0f, 0.0f), 7.0f), dyq = (uint32_t)fminf(fmaxf((r1.x * 0.5f + 0.5f) * 8.0f, 0.0f), 7.0f),
                           dzq = (uint32_t)fminf(fmaxf((r1.y * 0.5f + 0.5f) * 8.0f, 0.0f), 7.0f);
            const uint32_t m16 = spread3(cx >> 2) | (spread3(cy >> 2) << 1) | (spread3(cz >> 2) << 2);
            key = (m16 << 9) | (dxq << 6) | (dyq << 3) | dzq;
        } else if (L.sort_mode == 3) { // octant first, then the 64^3 cell
            key = (oct << 18) | morton;
        } else if (L.sort_mode == 4) { // 64^3 cell, octant, then which of the octant's 8 sub-cones the direction lies in (24 bits)
            const float ax = __builtin_fabsf(r0.w), ay = __builtin_fabsf(r1.x), az = __builtin_fabsf(r1.y);
            const uint32_t sub = (ax > ay ? 1u : 0u) | (ay > az ? 2u : 0u) | (ax > az ? 4u : 0u);
            key = (((morton << 3) | oct) << 3) | sub;
        } else if (L.sort_mode == 5) { // octant, 64^3 cell, then the sub-cone (24 bits)
            const float ax = __builtin_fabsf(r0.w), ay = __builtin_fabsf(r1.x), az = __builtin_fabsf(r1.y);
            const uint32_t sub = (ax > ay ? 1u : 0u) | (ay > az ? 2u : 0u) | (ax > az ? 4u : 0u);
            key = (oct << 21) | (morton << 3) | sub;
        } else if (L.sort_mode == 6) { // octant, 128^3 cell, sub-cone (27 bits): a finer origin cell for trees far beyond the caches
            const uint32_t fx7 = (uint32_t)fminf(fmaxf(fx * 128.0f, 0.0f), 127.0f), fy7 = (uint32_t)fminf(fmaxf(fy * 128.0f, 0.0f), 127.0f), fz7 = (uint32_t)fminf(fmaxf(fz * 128.0f, 0.0f), 127.0f);
            const uint32_t m21 = (morton << 3) | (fx7 & 1u) | ((fy7 & 1u) << 1) | ((fz7 & 1u) << 2); // the 64^3 code refined by one more bit per axis
            const float ax = __builtin_fabsf(r0.w), ay = __builtin_fabsf(r1.x), az = __builtin_fabsf(r1.y);
            const uint32_t sub = (ax > ay ? 1u : 0u) | (ay > az ? 2u : 0u) | (ax > az ? 4u : 0u);
            key = (oct << 24) | (m21 << 3) | sub;
        }
        L.sort_keys[0][j] = key;
        L.sort_vals[0][j] = val;
    }
}

// next bounce: the out queue becomes the in queue (the host swaps the pointers). The sub-queue fill counters turn into the
// dense prefix run_start[] the ray-order pass reads, and are cleared for the next wf_shade. One wave.
__global__ __launch_bounds__(64) void wf_advance(uint32_t *counters, uint32_t *stripes) {
    const uint32_t k = threadIdx.x; // == WF_STRIPES lanes
    const uint32_t c = stripes[k * WF_STRIPE_WORDS];
    uint32_t incl = c;
#pragma unroll
    for (uint32_t d = 1; d < WF_STRIPES; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (k >= d)
            incl += up;
    }
    uint32_t *run_start = stripes + WF_STRIPES * WF_STRIPE_WORDS;
    run_start[k] = incl - c;
    stripes[k * WF_STRIPE_WORDS] = 0u;
    if (k == WF_STRIPES - 1u) {
        run_start[WF_STRIPES] = incl;
        counters[WF_CNT_SLOTS] = (counters[WF_CNT_IN] + 63u) >> 6; // wave slots of the launch that just wrote the new in-queue
        counters[WF_CNT_IN] = incl;
        counters[WF_CNT_TICKET] = 0;
    }
    if (k < 8u)
        counters[WF_CNT_XCD + k * WF_CNT_XCD_STRIDE] = 0u;
}

// ------------------------------------------------------------------------------------------------ fold
// The return path of the recursion (raytracer.h:588-590): every finished path's pending frames `emission + inner * scl`, innermost
// first, then sanitize_nans (:607-616). One lane per path of the pass, all lanes busy; frames are stored level by level (WfLaunch::fold), so a
// wave reads 2 KB of consecutive records per level and nothing it does not need.
__global__ __launch_bounds__(256) void wf_fold(const WfLaunch L) {
    for (uint32_t path = blockIdx.x * blockDim.x + threadIdx.x; path < L.n_paths; path += gridDim.x * blockDim.x) {
        const RtF4 v = L.sample_out[path];
        V3 res = mk(v.x, v.y, v.z);
        uint32_t nb = __float_as_uint(v.w);
        while (nb > 0) {
            --nb;
            const float4 *fr = reinterpret_cast<const float4 *>(L.fold + ((size_t)nb * L.n_paths + path)); // frame level nb: adjacent lanes, adjacent records
            const float4 fe = fr[0], fs = fr[1];
            const V3 clr = res * mk(fs.x, fs.y, fs.z);
            res = mk(fe.x, fe.y, fe.z) + clr;
        }
        if (isnan_f(res.x))
            res.x = 0;
        if (isnan_f(res.y))
            res.y = 0;
        if (isnan_f(res.z))
            res.z = 0;
        L.sample_out[path] = RtF4{res.x, res.y, res.z, 0.f};
    }
}

// ------------------------------------------------------------------------------------------------ resolve
// render_pixel's `res += sanitize_nans(...)` loop (raytracer.h:621-625) for the samples of this pass, in sample order,
// continuing the running sum of earlier passes; the final pass divides by the sample count (:626).
__global__ __launch_bounds__(256) void wf_resolve(const WfLaunch L, int first_pass, int last_pass) {
    for (uint32_t lp = blockIdx.x * blockDim.x + threadIdx.x; lp < L.pass_pixels; lp += gridDim.x * blockDim.x) {
        V3 acc = mk(0, 0, 0);
        if (!first_pass) {
            const RtF4 a = L.accum[lp];
            acc = mk(a.x, a.y, a.z);
        }
        const RtF4 *src = L.sample_out + (size_t)lp * L.pass_samples;
        for (uint32_t ds = 0; ds < L.pass_samples; ++ds) {
            const RtF4 v = src[ds];
            acc = acc + mk(v.x, v.y, v.z);
        }
        if (last_pass) {
            const V3 out = acc / (float)L.samples;
            float *dst = L.fb + 3ull * wf_global_pixel(L, L.first_pixel + lp);
            dst[0] = out.x;
            dst[1] = out.y;
            dst[2] = out.z;
        } else {
            L.accum[lp] = RtF4{acc.x, acc.y, acc.z, 0.f};
        }
    }
}

// ------------------------------------------------------------------------------------------------ probe: rays in, hits out
// rt_cast_rays_ex: arbitrary rays go through the SAME closest-hit kernels the renderer launches. wf_from_rays writes them as
// queue records (what wf_generate / wf_shade write for their rays), wf_hits_out turns the hit records into the probe's output.
__global__ __launch_bounds__(256) void wf_from_rays(const WfLaunch L, const float *rays, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0)
        L.counters[WF_CNT_IN] = n;
    if (i >= n)
        return;
    const V3 o = ld3(rays + 6ull * i), d = ld3(rays + 6ull * i + 3);
    float4 *rq = reinterpret_cast<float4 *>(L.paths_in + i);
    rq[0] = make_float4(o.x, o.y, o.z, d.x);
    rq[1] = make_float4(d.y, d.z, __uint_as_float(i), __uint_as_float(1u));
    rq[2] = make_float4(1.0f / d.x, 1.0f / d.y, 1.0f / d.z, __uint_as_float(ray_fast_ok_ray(o, d) ? 1u : 0u));
    *reinterpret_cast<uint4 *>(rq + 3) = make_uint4(0u, 0u, 0u, 0u);
}
__global__ __launch_bounds__(256) void wf_hits_out(const DevScene S, const WfLaunch L, uint32_t n, uint32_t *prim_out, float *bct_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float4 hq = *reinterpret_cast<const float4 *>(L.hits + i);
    const uint32_t k = __float_as_uint(hq.x);
    if (k == RT_NONE) {
        prim_out[i] = RT_NONE;
        bct_out[3ull * i] = bct_out[3ull * i + 1] = bct_out[3ull * i + 2] = 0.0f;
    } else {
        prim_out[i] = (k & RT_PRIM_FLAG) ? S.n_triangles + (k & ~RT_PRIM_FLAG) : S.scene.tris[k].prim;
        bct_out[3ull * i] = hq.y;
        bct_out[3ull * i + 1] = hq.z;
        bct_out[3ull * i + 2] = hq.w;
    }
}

} // namespace

namespace rt {

// One pass of the pipeline, fully stream-ordered (no host synchronisation): generate, ray_depth x (extend, shade,
// advance), resolve. `L.paths_in/paths_out` are swapped locally per bounce.
// every launch is checked: a failed launch (bad configuration, lost device) must not turn into a silently wrong image
#define WF_LAUNCH(...)                                  \
    do {                                                \
        if (hipError_t le_ = RT_LAUNCH_CHECKED(__VA_ARGS__); le_ != hipSuccess) \
            return le_;                                 \
    } while (0)

// the closest-hit kernel of one bounce: packets for coherent primary rays or the per-lane persistent kernel, each in the
// reference's traversal order (near-local pruning, the parity mode) or with global-best pruning (L.global_best, production)
static hipError_t launch_extend(const DevScene &S, const WfLaunch &L, bool packet, bool stats, int ext_blocks, hipStream_t stream) {
    if (S.scene.wide) // production build (RT_BUILD_WIDE): every bounce walks the 8-wide tree (rt_wide.hip)
        return launch_extend_wide(S, L, packet, stats, ext_blocks, stream);
    const dim3 grid(ext_blocks), block(256);
    const bool gb = L.global_best != 0u;
#define EXT_CASE(P, ST, G)                                                                          \
    if (packet == P && stats == ST && gb == G) {                                                    \
        if (P)                                                                                      \
            return RT_LAUNCH_CHECKED((wf_extend_packet<ST, G>), grid, block, 0, stream, S, L);      \
        return RT_LAUNCH_CHECKED((wf_extend<ST, G>), grid, block, 0, stream, S, L);                 \
    }
    EXT_CASE(false, false, false)
    EXT_CASE(false, false, true)
    EXT_CASE(false, true, false)
    EXT_CASE(false, true, true)
    EXT_CASE(true, false, false)
    EXT_CASE(true, false, true)
    EXT_CASE(true, true, false)
    EXT_CASE(true, true, true)
#undef EXT_CASE
    return hipErrorInvalidValue;
}

// ---- closest-hit probe through the production kernels (rt_cast_rays_ex): rays -> queue -> wf_extend / wf_extend_packet -> hits
hipError_t launch_wavefront_cast(const DevScene &S, WfLaunch L, const float *rays, uint32_t n, bool packet, bool stats, uint32_t *prim, float *bct,
                                 hipStream_t stream) {
    const dim3 block(256);
    hipError_t e = hipMemsetAsync(L.counters, 0, sizeof(uint32_t) * WF_CNT_WORDS, stream);
    if (e != hipSuccess)
        return e;
    L.n_paths = n;
    L.order = nullptr, L.order_classed = 0u;
    L.packet_census = nullptr;
    const int blocks = (int)((n + 255u) / 256u);
    WF_LAUNCH(wf_from_rays, dim3(blocks), block, 0, stream, L, rays, n);
    if ((e = launch_extend(S, L, packet, stats, (int)(L.stack_stride / 256u), stream)) != hipSuccess)
        return e;
    if (S.n_prims)
        WF_LAUNCH(wf_extend_prims, dim3(blocks), block, 0, stream, S, L);
    WF_LAUNCH(wf_hits_out, dim3(blocks), block, 0, stream, S, L, n, prim, bct);
    return hipSuccess;
}

hipError_t launch_wavefront_pass(const DevScene &S, WfLaunch L, bool stats, int num_cus, bool first_pass, bool last_pass, hipStream_t stream,
                                 EventPool *extend_events, unsigned long long *packet_census_out, const WfHostSync *host_sync) {
    const int gen_blocks = (int)((L.n_paths + 255u) / 256u < (uint32_t)num_cus * 16u ? (L.n_paths + 255u) / 256u : (uint32_t)num_cus * 16u);
    const dim3 block(256);
    hipError_t e = hipMemsetAsync(L.counters, 0, sizeof(uint32_t) * WF_CNT_WORDS, stream);
    if (e != hipSuccess)
        return e;
    if ((e = hipMemsetAsync(L.stripes, 0, sizeof(uint32_t) * WF_STRIPE_BUF_WORDS, stream)) != hipSuccess)
        return e;
    const bool packet = L.use_packet != 0u && L.packet_census != nullptr;
    if (packet && (e = hipMemsetAsync(L.packet_census, 0, 2 * sizeof(unsigned long long), stream)) != hipSuccess)
        return e;
    if (packet_census_out)
        packet_census_out[0] = packet_census_out[1] = 0ull;
    if (stats)
        WF_LAUNCH((wf_generate<true>), dim3(gen_blocks > 0 ? gen_blocks : 1), block, 0, stream, S, L);
    else
        WF_LAUNCH((wf_generate<false>), dim3(gen_blocks > 0 ? gen_blocks : 1), block, 0, stream, S, L);
    const int ext_blocks = (int)(L.stack_stride / 256u); // rt_scene.cpp sizes the overflow workspace for exactly this grid
    const int shade_blocks = num_cus * RT_SHADE_BLOCKS_PER_CU;
    // Queue sizes reach the host one bounce LATE and without ever idling the device: after bounce b's wf_advance the size of
    // queue b + 1 is copied to pinned word b + 1 and an event is recorded; bounce b + 2 waits for THAT event — by then bounce
    // b + 1's kernels are queued behind it, so the device has a whole bounce of work while the host looks. The host needs the
    // size only as an upper bound (grid of the ray-order pass, length of the sort, "nothing left": stop): the kernels read the
    // exact size from device memory. (Round 2 synchronised the stream once per bounce: eight idle gaps per pass.)
    const WfHostSync *hs = L.sort_mode != 0u || (packet && packet_census_out) ? host_sync : nullptr;
    if (hs && (!hs->counts || !hs->events || hs->n_events < (int)L.ray_depth + 1))
        hs = nullptr;
    // the per-bounce size read-back (one host wait per bounce from bounce 2 on) is only worth its latency where a sort follows; an unsorted
    // pass (a small one, rt_scene.cpp; or RT_SORT_OFF) is queued in one go and only the packet census, if any, is waited for once
    const bool want_bound = hs && L.sort_mode != 0u;
    uint32_t bound = L.n_paths; // upper bound of the queue entering the bounce about to be launched
    bool census_pending = false; // the packet census of bounce 0 is on its way to the pinned words (event 0)
    for (uint32_t b = 0; b < L.ray_depth; ++b) {
        L.order = nullptr, L.order_classed = 0u; // primary rays: dense and coherent as generated
        if (b > 0) {
            if (census_pending && b == 2) { // bounce 0's census: its copy is two bounces behind the queue head by now
                if ((e = hipEventSynchronize(hs->events[0])) != hipSuccess)
                    return e;
                std::memcpy(packet_census_out, hs->counts + WF_HOST_CENSUS_WORD, 2 * sizeof(unsigned long long));
                census_pending = false;
            }
            if (want_bound && b >= 2) { // size of queue b - 1, an upper bound of queue b
                if ((e = hipEventSynchronize(hs->events[b - 1])) != hipSuccess)
                    return e;
                bound = hs->counts[b - 1];
                if (bound == 0)
                    break;
            }
            const bool sort = want_bound && bound >= 4096u;
            // dense ray index -> slot of paths_in (wf_shade's sub-queue regions), with the coherence keys when a sort follows
            const uint32_t kb = (bound + 255u) / 256u < (uint32_t)num_cus * 16u ? (bound + 255u) / 256u : (uint32_t)num_cus * 16u;
            WF_LAUNCH(wf_sort_keys, dim3(kb > 0 ? kb : 1), block, 0, stream, S, L, sort ? 0 : 1, bound);
            if (sort) {
                size_t tmp = L.sort_temp_bytes;
                hipError_t se = rocprim::radix_sort_pairs(L.sort_temp, tmp, L.sort_keys[0], L.sort_keys[1], L.sort_vals[0], L.sort_vals[1], (size_t)bound, 0u, L.sort_mode == 6 ? 27u : L.sort_mode >= 4 ? 24u : 21u, stream);
                if (se != hipSuccess)
                    return se;
            }
            L.order = L.sort_vals[1];
            L.order_classed = sort ? 1u : 0u;
        }
        // time the dominant kernel per launch (bench.py roofline): HIP events on the launch stream, taken from the scene's
        // pool (created once, reused by every render)
        hipEvent_t e0 = extend_events ? extend_events->next() : nullptr;
        hipEvent_t e1 = e0 ? extend_events->next() : nullptr;
        if (e0 && e1)
            (void)hipEventRecord(e0, stream);
        if (hipError_t xe = launch_extend(S, L, b == 0 && packet, stats, ext_blocks, stream); xe != hipSuccess)
            return xe;
        if (e0 && e1)
            (void)hipEventRecord(e1, stream);
        if (S.n_prims)
            WF_LAUNCH(wf_extend_prims, dim3(shade_blocks), block, 0, stream, S, L);
        // ENV: the scene has an environment map (DevScene::bg_tex): the miss branch looks it up (scene.h:83-89); those instantiations read the
        // light tables from global memory (LIGHTS_LDS only saves latency), so a scene without one never pays for the lookup's registers
        const bool env = S.bg_tex >= 0;
        const bool lights_lds = S.lights.lds_inner != 0u && !env;
        if (env && stats)
            WF_LAUNCH((wf_shade<true, false, true>), dim3(shade_blocks), block, 0, stream, S, L);
        else if (env)
            WF_LAUNCH((wf_shade<false, false, true>), dim3(shade_blocks), block, 0, stream, S, L);
        else if (stats && lights_lds)
            WF_LAUNCH((wf_shade<true, true, false>), dim3(shade_blocks), block, 0, stream, S, L);
        else if (stats)
            WF_LAUNCH((wf_shade<true, false, false>), dim3(shade_blocks), block, 0, stream, S, L);
        else if (lights_lds)
            WF_LAUNCH((wf_shade<false, true, false>), dim3(shade_blocks), block, 0, stream, S, L);
        else
            WF_LAUNCH((wf_shade<false, false, false>), dim3(shade_blocks), block, 0, stream, S, L);
        WF_LAUNCH(wf_advance, dim3(1), dim3(64), 0, stream, L.counters, L.stripes);
        if (hs && b == 0 && packet && packet_census_out) { // the packet kernel's census -> pinned words; read at bounce 2, or behind the loop
            if ((e = hipMemcpyAsync(hs->counts + WF_HOST_CENSUS_WORD, L.packet_census, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream)) != hipSuccess)
                return e;
            if ((e = hipEventRecord(hs->events[0], stream)) != hipSuccess)
                return e;
            census_pending = true;
        }
        if (want_bound && b + 1 < L.ray_depth) { // size of queue b + 1 -> pinned word b + 1 (read by bounce b + 2)
            if ((e = hipMemcpyAsync(hs->counts + b + 1, L.counters + WF_CNT_IN, sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess)
                return e;
            if ((e = hipEventRecord(hs->events[b + 1], stream)) != hipSuccess)
                return e;
        }
        WfPath *t = L.paths_in;
        L.paths_in = L.paths_out;
        L.paths_out = t;
    }
    WF_LAUNCH(wf_fold, dim3(gen_blocks > 0 ? gen_blocks : 1), block, 0, stream, L);
    if (census_pending) { // ray_depth <= 2 (no bounce 2 to read it at): wait for bounce 0 only; the rest of the pass is queued behind it
        if ((e = hipEventSynchronize(hs->events[0])) != hipSuccess)
            return e;
        std::memcpy(packet_census_out, hs->counts + WF_HOST_CENSUS_WORD, 2 * sizeof(unsigned long long));
    }
    const int res_blocks = (int)((L.pass_pixels + 255u) / 256u);
    WF_LAUNCH(wf_resolve, dim3(res_blocks > 0 ? res_blocks : 1), block, 0, stream, L, first_pass ? 1 : 0, last_pass ? 1 : 0);
    return hipSuccess;
}

size_t wavefront_sort_temp_bytes(size_t n) {
    size_t tmp = 0;
    uint32_t *k = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, k, k, k, k, n, 0u, 27u, (hipStream_t) nullptr);
    return tmp;
}

} // namespace rt

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
#include "rt_device_lib.h"
#include "rt_kernels.h"

namespace {

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

// Lane states of the persistent loop.
//   ST_IDLE  : no work item (needs a refill or the image is exhausted)
//   ST_NEW   : owns a pixel, next sample not started
//   ST_TRAV  : a closest-hit traversal is in flight (resumable, see trav_step)
//   ST_READY : traversal finished, hit record waits to be shaded
enum { ST_IDLE = 0, ST_NEW = 1, ST_TRAV = 2, ST_READY = 3 };

// When fewer than TRAV_MIN_LANES lanes of the wave are still traversing and others wait to be shaded, the traversal
// loop is left: the finished lanes shade, generate their next ray and re-enter together with the stragglers, which
// resume where they stopped. Keeps the 64-wide wave dense through the heavy-tailed traversal lengths.
#ifndef RT_TRAV_MIN_LANES
#define RT_TRAV_MIN_LANES 40
#endif
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 4 /* <= 128 VGPRs: 4 waves/SIMD measured fastest (tools_sweep.sh) */
#endif
constexpr int TRAV_MIN_LANES = RT_TRAV_MIN_LANES;

template <int MODE, bool STATS>
__global__ __launch_bounds__(256, RT_WAVES_PER_SIMD) void render_kernel(const DevScene S, const RenderLaunch L) {
    __shared__ float s_lin[256];
    __shared__ float s_gam[256];
    __shared__ uint32_t s_stack[STACK_LDS_DWORDS];
    s_lin[threadIdx.x] = S.lut_linear[threadIdx.x];
    s_gam[threadIdx.x] = S.lut_gamma[threadIdx.x];
    __syncthreads();

#ifdef RT_DIAG
    if (STATS && threadIdx.x == 0 && blockIdx.x == 0)
        g_diag = L.stats;
#endif
    LaneStats<STATS> st;
    Rng<MODE> rng;
    RT_DECLARE_STACK(stk, LDS_DEPTH, s_stack);
    float fold_e[RT_MAX_RAY_DEPTH * 3];
    float fold_s[RT_MAX_RAY_DEPTH * 3];

    const V3 cam_pos = ld3(S.cam_pos), cam_right = ld3(S.cam_right), cam_up = ld3(S.cam_up), cam_fwd = ld3(S.cam_fwd);
    const bool has_lights = S.lights.n_tris != 0; // raytracer.h:449-453

    int state = ST_IDLE;
    bool exhausted = false;
    uint32_t pix = 0, pix_end = 0, s = 0, depth_left = 0, nb = 0;
    V3 acc{0, 0, 0};
    Trav T;
    T.cur = T_DONE;
    T.sp = 0;

    for (;;) {
        // ---- refill idle lanes: ballot + prefix count, one ticket atomic per wave
        if (state == ST_IDLE && !exhausted) {
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
                state = pix < pix_end ? ST_NEW : ST_IDLE;
                if constexpr (MODE == RT_RNG_REFERENCE)
                    rt_minstd_seed(&rng.g, it.seed); // RaytracerThreadContext(ctx, span) :648
            } else {
                exhausted = true;
            }
        }
        if (__ballot(state != ST_IDLE) == 0ull)
            break;

        // ---- render_pixel loop head :621-622 + gen_ray :527-538
        if (state == ST_NEW) {
            if constexpr (MODE == RT_RNG_DEVICE)
                rt_xoshiro_seed(&rng.g, L.seed, pix, s);
            const uint32_t x = pix % L.width, y = pix / L.width;
            float ox = uniform_real(rng, 0.0f, 1.0f);
            float oy = uniform_real(rng, 0.0f, 1.0f);
            float sx = (2 * ((float)(int)x + ox) / (float)L.width - 1) * L.tan_x;
            float sy = (2 * ((float)(int)y + oy) / (float)L.height - 1) * L.tan_y;
            V3 rd = norm(sx * cam_right - sy * cam_up + 1.0f * cam_fwd);
            depth_left = S.ray_depth;
            nb = 0;
            st.cast();
            trav_init(T, S.scene, cam_pos, rd); // ray_depth >= 1 here, so trace_ray casts (:600)
            state = T.cur == T_DONE ? ST_READY : ST_TRAV;
        }

        // ---- closest-hit traversal, one record per lane per step
        for (;;) {
            const unsigned long long tm = __ballot(state == ST_TRAV);
            if (tm == 0ull)
                break;
            if (__popcll(tm) < TRAV_MIN_LANES && __ballot(state == ST_READY) != 0ull)
                break;
            DIAG(0, 1);
            DIAG(1, (unsigned long long)__popcll(tm));
            if (state == ST_TRAV) {
                trav_step<STATS>(T, S.scene, stk, EPS, st);
                if (T.cur == T_DONE)
                    state = ST_READY;
            }
        }

        // ---- trace_ray's hit / miss branch (:602-604) and shade (:555-591) for the lanes whose traversal finished
        DIAG(11, 1);
        if (state == ST_READY) {
            DIAG(9, 1);
            DIAG_LANES(10);
            Hit h = T.best;
            if (S.n_prims)
                prims_closest(S, T.o, T.d, h);
            if (h.k != RT_NONE)
                depth_left -= 1; // shade(..., max_depth - 1)
            const ShadeResult sr = shade_hit<Rng<MODE>, STATS>(S, light_tabs_global(S), h, T.o, T.d, rng, has_lights, stk, s_lin, s_gam, st);
            bool terminal = sr.terminal;
            V3 term = sr.term;
            const V3 nro = sr.nro, nrd = sr.nrd;
            if (sr.push) { // emission + trace_ray(...) * scl  (:588-590) deferred to the fold below
                fold_e[3 * nb + 0] = sr.emission.x;
                fold_e[3 * nb + 1] = sr.emission.y;
                fold_e[3 * nb + 2] = sr.emission.z;
                fold_s[3 * nb + 0] = sr.scl.x;
                fold_s[3 * nb + 1] = sr.scl.y;
                fold_s[3 * nb + 2] = sr.scl.z;
                ++nb;
            }
            if (!terminal) {
                if (depth_left == 0) { // trace_ray(..., 0) returns (0,0,0) without casting (:596-598)
                    terminal = true;
                    term = mk(0, 0, 0);
                } else {
                    st.cast();
                    trav_init(T, S.scene, nro, nrd);
                    state = T.cur == T_DONE ? ST_READY : ST_TRAV;
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
                state = ST_NEW;
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
                        state = ST_IDLE;
                }
            }
        }
    }
#ifndef RT_DIAG
    st.flush(L.stats);
#endif
}

// ---------------------------------------------------------------------------------------------- probe kernels
__global__ __launch_bounds__(256) void cast_kernel(const DevScene S, const float *rays, uint32_t n, uint32_t *prim_out, float *bct_out) {
    __shared__ uint32_t s_stack[STACK_LDS_DWORDS];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    RT_DECLARE_STACK(stk, LDS_DEPTH, s_stack);
    LaneStats<false> st;
    Trav T;
    trav_init(T, S.scene, ld3(rays + 6ull * i), ld3(rays + 6ull * i + 3));
    while (T.cur != T_DONE)
        trav_step<false>(T, S.scene, stk, EPS, st);
    Hit h = T.best;
    if (S.n_prims)
        prims_closest(S, T.o, T.d, h);
    if (h.k == RT_NONE) {
        prim_out[i] = RT_NONE;
        bct_out[3ull * i] = bct_out[3ull * i + 1] = bct_out[3ull * i + 2] = 0.0f;
    } else {
        prim_out[i] = (h.k & RT_PRIM_FLAG) ? S.n_triangles + (h.k & ~RT_PRIM_FLAG) : S.scene.tris[h.k].prim;
        bct_out[3ull * i] = h.b;
        bct_out[3ull * i + 1] = h.c;
        bct_out[3ull * i + 2] = h.t;
    }
}

// rt_surface_normals: cast_kernel's closest hit, then the normals make_surf (to_intersection_info, bvh.h:80-121) gives shade()
__global__ __launch_bounds__(256) void surface_normals_kernel(const DevScene S, const float *rays, uint32_t n, uint32_t *prim_out, float *t_out, float *normal_out, float *shading_out) {
    __shared__ uint32_t s_stack[STACK_LDS_DWORDS];
    __shared__ float s_lin[256];
    __shared__ float s_gam[256];
    s_lin[threadIdx.x] = S.lut_linear[threadIdx.x];
    s_gam[threadIdx.x] = S.lut_gamma[threadIdx.x];
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    RT_DECLARE_STACK(stk, LDS_DEPTH, s_stack);
    LaneStats<false> st;
    Trav T;
    trav_init(T, S.scene, ld3(rays + 6ull * i), ld3(rays + 6ull * i + 3));
    while (T.cur != T_DONE)
        trav_step<false>(T, S.scene, stk, EPS, st);
    Hit h = T.best;
    if (S.n_prims)
        prims_closest(S, T.o, T.d, h);
    V3 nn = mk(0.f, 0.f, 0.f), sn = nn;
    if (h.k != RT_NONE) {
        const Surf s = make_surf<false>(S, h, T.o, T.d, s_lin, s_gam, st);
        nn = s.normal;
        sn = s.shading_normal;
    }
    prim_out[i] = h.k == RT_NONE ? RT_NONE : (h.k & RT_PRIM_FLAG) ? S.n_triangles + (h.k & ~RT_PRIM_FLAG) : S.scene.tris[h.k].prim;
    t_out[i] = h.k == RT_NONE ? 0.0f : h.t;
    if (normal_out)
        normal_out[3ull * i] = nn.x, normal_out[3ull * i + 1] = nn.y, normal_out[3ull * i + 2] = nn.z;
    if (shading_out)
        shading_out[3ull * i] = sn.x, shading_out[3ull * i + 1] = sn.y, shading_out[3ull * i + 2] = sn.z;
}

__global__ __launch_bounds__(256) void light_pdf_kernel(const DevScene S, const float *rays, uint32_t n, float *pdf_out) {
    __shared__ uint32_t s_stack[STACK_LDS_DWORDS];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    RT_DECLARE_STACK(stk, LDS_DEPTH, s_stack);
    LaneStats<false> st;
    V3 o = ld3(rays + 6ull * i), d = ld3(rays + 6ull * i + 3);
    pdf_out[i] = S.lights.n_tris ? lights_pdf<false>(S, light_tabs_global(S), o, d, stk, st) : 0.0f;
}

// Scene::bg_at for explicit directions (rt_bg_at: the environment lookup on its own, for the parity tests)
__global__ __launch_bounds__(256) void bg_at_kernel(const DevScene S, const float *dirs, uint32_t n, float *rgb_out) {
    __shared__ float s_lin[256];
    __shared__ float s_gam[256];
    s_lin[threadIdx.x] = S.lut_linear[threadIdx.x];
    s_gam[threadIdx.x] = S.lut_gamma[threadIdx.x];
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    LaneStats<false> st;
    const V3 c = bg_at<false>(S, ld3(dirs + 3ull * i), s_lin, s_gam, st);
    rgb_out[3ull * i] = c.x;
    rgb_out[3ull * i + 1] = c.y;
    rgb_out[3ull * i + 2] = c.z;
}

} // namespace


namespace rt {

hipError_t launch_bg_at(const DevScene &S, const float *dirs, uint32_t n, float *rgb, hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    return RT_LAUNCH_CHECKED(bg_at_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, S, dirs, n, rgb);
}

hipError_t launch_render(const DevScene &S, const RenderLaunch &L, bool stats, int blocks, hipStream_t stream) {
    dim3 grid(blocks), block(256);
    if (L.rng_mode == RT_RNG_REFERENCE) {
        if (stats)
            return RT_LAUNCH_CHECKED((render_kernel<RT_RNG_REFERENCE, true>), grid, block, 0, stream, S, L);
        return RT_LAUNCH_CHECKED((render_kernel<RT_RNG_REFERENCE, false>), grid, block, 0, stream, S, L);
    }
    if (stats)
        return RT_LAUNCH_CHECKED((render_kernel<RT_RNG_DEVICE, true>), grid, block, 0, stream, S, L);
    return RT_LAUNCH_CHECKED((render_kernel<RT_RNG_DEVICE, false>), grid, block, 0, stream, S, L);
}

hipError_t launch_cast(const DevScene &S, const float *rays, uint32_t n, uint32_t *prim, float *bct, hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    return RT_LAUNCH_CHECKED(cast_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, S, rays, n, prim, bct);
}

hipError_t launch_surface_normals(const DevScene &S, const float *rays, uint32_t n, uint32_t *prim, float *t, float *normal, float *shading, hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    return RT_LAUNCH_CHECKED(surface_normals_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, S, rays, n, prim, t, normal, shading);
}

hipError_t launch_light_pdf(const DevScene &S, const float *rays, uint32_t n, float *pdf, hipStream_t stream) {
    if (n == 0)
        return hipSuccess;
    return RT_LAUNCH_CHECKED(light_pdf_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, S, rays, n, pdf);
}

} // namespace rt

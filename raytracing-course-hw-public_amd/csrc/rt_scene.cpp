// rt_scene.cpp — C-ABI entry points of include/rt_abi.h: scene residency + launches.
//
// rt_create  replaces RaytracerStaticContext(scene) (src/raytracer.h:440-454): two host BVH builds in the
//            reference topology, flattening, upload to HBM.
// rt_render  replaces run_raytracer(scene, image) (src/raytracer.h:629-674): one persistent-wavefront launch.
// There is no CPU fallback anywhere in this file: without a HIP device every device entry point fails.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "bvh_build.h"
#include "rt_bvh_device.h"
#include "rt_device_types.h"
#include "rt_error.h"
#include "rt_film.h"
#include "rt_group.h"
#include "rt_kernels.h"
#include "wide_build.h"

namespace {

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return rt::fail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : (e_ == hipErrorNoDevice ? RT_ERR_NO_DEVICE : RT_ERR_HIP), \
                            std::string(#expr) + ": " + hipGetErrorString(e_));                           \
    } while (0)

template <class T> int upload(const std::vector<T> &v, const T **out, std::vector<void *> &owned) {
    *out = nullptr;
    if (v.empty())
        return RT_OK;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
    owned.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T *>(p);
    return RT_OK;
}

// wf_advance scans the sub-queue fill levels with ONE wave (a lane per sub-queue) and wf_sort_keys finds a ray's sub-queue with
// a power-of-two binary search (rt_wavefront.hip)
#ifndef RT_WIDE_PACKET_MIN_LANES
#define RT_WIDE_PACKET_MIN_LANES 20.0 /* the wide packet kernel wins from 27 lanes per trip (4 SPP per pass) upwards: tools/packet_calibration.py */
#endif
static_assert(WF_STRIPES == 64u && (WF_STRIPES & (WF_STRIPES - 1u)) == 0u, "WF_STRIPES must equal the wave size (64)");

// device allocation that is released on every return path of the probe entry points
struct DevBuf {
    void *p = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T> T *as() const { return static_cast<T *>(p); }
    ~DevBuf() {
        if (p)
            (void)hipFree(p);
    }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

struct V3h {
    float x, y, z;
};
inline V3h cross_h(V3h a, V3h b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float len_h(V3h a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

} // namespace

namespace rt {
struct PreparedScene;
}

struct rt_scene {
    rt::Group *group = nullptr; // multi-GPU scene: replicas + RCCL communicator (rt_group.cpp); the fields below stay unused
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    rt::EventPool ext_events; // (start, stop) per wf_extend launch of the current render; reused by every render
    DevScene dev{};
    rt_camera cam{};
    std::vector<void *> owned;
    std::shared_ptr<const rt::PreparedScene> prep; // the host half of rt_create, shared by the replicas of a multi-GPU scene
    rt::HostBvh rebuilt_bvh;   // device-built scene BVH: the reference-style node list, reconstructed from HBM on demand
    bool device_built = false; // scene BVH built by rt_bvh_device.hip
    bool wide_built = false;   // RT_BUILD_WIDE: the scene BVH in HBM is the 8-wide quantised tree (wide_build.cpp); host_bvh[0] is the
                               // binary tree it was collapsed from when that one was built on the host
    uint32_t wide_depth = 0;
    double wide_ms = 0, wide_cost = 0;
    uint32_t dev_n_inner[2] = {0, 0};
    double build_ms = 0, build_upload_ms = 0;
    uint32_t *d_counter = nullptr;
    DevStats *d_stats = nullptr;
    float *d_fb = nullptr;
    size_t fb_capacity = 0; // floats
    uint8_t *d_rgb8 = nullptr; // device film output (rt_render_rgb8 with a host destination, rt_film_rgb8)
    size_t rgb8_capacity = 0;
    rt::FilmTable *d_film_table = nullptr;

    int ensure_fb(size_t fb_floats) {
        if (fb_capacity >= fb_floats)
            return RT_OK;
        if (d_fb)
            (void)hipFree(d_fb);
        d_fb = nullptr;
        fb_capacity = 0;
        void *q = nullptr;
        HIP_TRY(hipMalloc(&q, fb_floats * sizeof(float)));
        d_fb = static_cast<float *>(q);
        fb_capacity = fb_floats;
        return RT_OK;
    }
    // device film prerequisites: the verified threshold table (host/film.cpp) and, optionally, an rgb8 staging buffer
    int ensure_film(size_t rgb8_bytes) {
        if (!d_film_table) {
            rt::FilmTable t{};
            if (!rt::film_table(t.thr, t.special))
                return rt::fail(RT_ERR_UNSUPPORTED, "device film: the host libm's powf failed the monotonicity check; use rt_render + rt_tonemap_rgb8");
            void *q = nullptr;
            HIP_TRY(hipMalloc(&q, sizeof(t)));
            d_film_table = static_cast<rt::FilmTable *>(q);
            HIP_TRY(hipMemcpyAsync(d_film_table, &t, sizeof(t), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream)); // `t` is a local
        }
        if (rgb8_capacity < rgb8_bytes) {
            if (d_rgb8)
                (void)hipFree(d_rgb8);
            d_rgb8 = nullptr;
            rgb8_capacity = 0;
            void *q = nullptr;
            HIP_TRY(hipMalloc(&q, rgb8_bytes));
            d_rgb8 = static_cast<uint8_t *>(q);
            rgb8_capacity = rgb8_bytes;
        }
        return RT_OK;
    }
    int num_cus = 0;
    int blocks_per_cu = 8; // upper bound on resident 256-thread blocks per CU; surplus blocks find the ticket exhausted
    // wavefront pipeline workspace (rt_wavefront.hip), sized for wf_paths_cap paths / wf_pixels_cap pixels per pass
    uint64_t wf_paths_cap = 0, wf_pixels_cap = 0;
    uint32_t wf_depth_cap = 0;
    std::vector<void *> wf_owned;
    WfPath *wf_paths[2] = {nullptr, nullptr};
    uint32_t *wf_stripes = nullptr;
    WfHit *wf_hits = nullptr;
    WfFold *wf_fold = nullptr;
    RtF4 *wf_samples = nullptr, *wf_accum = nullptr;
    uint32_t *wf_counters = nullptr;
    void *wf_stack_overflow = nullptr; // wf_extend's evicted stack frames (RingStackT): RT_MAX_STACK x grid threads x 16 B
    uint32_t wf_stack_stride = 0;
    uint32_t *wf_sort_keys[2] = {nullptr, nullptr}, *wf_sort_vals[2] = {nullptr, nullptr};
    void *wf_sort_temp = nullptr;
    size_t wf_sort_temp_bytes = 0;
    uint32_t *wf_host_count = nullptr; // pinned, 48 words: queue size per bounce, then wf_extend_packet's census (rt_kernels.h WfHostSync)
    std::vector<hipEvent_t> wf_count_events; // one per bounce: "the size of the queue entering this bounce has reached wf_host_count"
    // wf_extend_packet (primary rays as coherent packets) pays off only while a wave's 64 rays stay together; its own census
    // (lanes served per trip) decides per configuration whether later passes and renders keep using it
    uint64_t pkt_key = 0; // width, height, samples per pass, shard count of the configuration pkt_off was measured on
    bool pkt_off = false;
    uint32_t pkt_lanes_x100 = 0; // last packet census: lanes served per trip x 100 (rt_stats.reserved)

    int ensure_wavefront(uint64_t paths, uint64_t pixels, uint32_t depth) {
        if (paths <= wf_paths_cap && pixels <= wf_pixels_cap && depth <= wf_depth_cap)
            return RT_OK;
        for (void *p : wf_owned)
            (void)hipFree(p);
        wf_owned.clear();
        wf_paths_cap = wf_pixels_cap = 0;
        wf_depth_cap = 0;
        auto alloc = [&](size_t bytes, void **out) -> int {
            *out = nullptr;
            hipError_t e = hipMalloc(out, bytes ? bytes : 16);
            if (e != hipSuccess)
                return rt::fail(e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, std::string("wavefront workspace: ") + hipGetErrorString(e));
            wf_owned.push_back(*out);
            return RT_OK;
        };
        int rc;
        // + 64: wf_shade's sub-queue regions cover whole wave slots (rt_device_types.h, WF_STRIPES)
        if ((rc = alloc((paths + 64) * sizeof(WfPath), (void **)&wf_paths[0])) != RT_OK || (rc = alloc((paths + 64) * sizeof(WfPath), (void **)&wf_paths[1])) != RT_OK ||
            (rc = alloc(WF_STRIPE_BUF_WORDS * sizeof(uint32_t), (void **)&wf_stripes)) != RT_OK ||
            (rc = alloc(paths * sizeof(WfHit), (void **)&wf_hits)) != RT_OK || (rc = alloc(paths * depth * sizeof(WfFold), (void **)&wf_fold)) != RT_OK ||
            (rc = alloc(paths * sizeof(RtF4), (void **)&wf_samples)) != RT_OK || (rc = alloc(pixels * sizeof(RtF4), (void **)&wf_accum)) != RT_OK ||
            (rc = alloc(WF_CNT_WORDS * sizeof(uint32_t) + 1024, (void **)&wf_counters)) != RT_OK)
            return rc;
        wf_stack_stride = (uint32_t)num_cus * 8u * 256u; // the wf_extend grid: 8 blocks of 256 threads per CU
        if ((rc = alloc((size_t)RT_MAX_STACK * wf_stack_stride * 16, &wf_stack_overflow)) != RT_OK)
            return rc;
        wf_sort_temp_bytes = rt::wavefront_sort_temp_bytes(paths);
        if ((rc = alloc(paths * 4, (void **)&wf_sort_keys[0])) != RT_OK || (rc = alloc(paths * 4, (void **)&wf_sort_keys[1])) != RT_OK ||
            (rc = alloc(paths * 4, (void **)&wf_sort_vals[0])) != RT_OK || (rc = alloc(paths * 4, (void **)&wf_sort_vals[1])) != RT_OK ||
            (rc = alloc(wf_sort_temp_bytes, &wf_sort_temp)) != RT_OK)
            return rc;
        // on the scene's own (non-blocking) stream: a null-stream memset is not ordered with the kernels launched there and
        // could land after wf_generate had set the queue size
        HIP_TRY(hipMemsetAsync(wf_counters, 0, WF_CNT_WORDS * sizeof(uint32_t) + 1024, stream));
        wf_paths_cap = paths;
        wf_pixels_cap = pixels;
        wf_depth_cap = depth;
        return RT_OK;
    }

    // the workspace half of a WfLaunch (after ensure_wavefront): queues, hit records, counters, stack workspace, sort buffers
    void wf_bind(WfLaunch &W) {
        W.paths_in = wf_paths[0];
        W.paths_out = wf_paths[1];
        W.hits = wf_hits;
        W.fold = wf_fold;
        W.sample_out = wf_samples;
        W.accum = wf_accum;
        W.counters = wf_counters;
        W.stripes = wf_stripes;
        W.stack_overflow = wf_stack_overflow;
        W.stack_stride = wf_stack_stride;
        W.diag = wf_counters + 64; // dev census words live behind the queue counters
        for (int k = 0; k < 2; ++k) { // sort_vals[1] also carries the unsorted order (RT_WF_SORT=0: sort_mode 0)
            W.sort_keys[k] = wf_sort_keys[k];
            W.sort_vals[k] = wf_sort_vals[k];
        }
        W.sort_temp = wf_sort_temp;
        W.sort_temp_bytes = wf_sort_temp_bytes;
        W.packet_census = reinterpret_cast<unsigned long long *>(wf_counters + 32);
    }

    ~rt_scene() {
        if (group) {
            rt::group_destroy(group);
            return;
        }
        (void)hipSetDevice(device);
        if (wf_host_count)
            (void)hipHostFree(wf_host_count);
        for (hipEvent_t ev : wf_count_events)
            (void)hipEventDestroy(ev);
        for (void *p : wf_owned)
            (void)hipFree(p);
        for (void *p : owned)
            (void)hipFree(p);
        if (d_fb)
            (void)hipFree(d_fb);
        if (d_rgb8)
            (void)hipFree(d_rgb8);
        if (d_film_table)
            (void)hipFree(d_film_table);
        ext_events.destroy();
        if (ev0)
            (void)hipEventDestroy(ev0);
        if (ev1)
            (void)hipEventDestroy(ev1);
        if (stream)
            (void)hipStreamDestroy(stream);
    }
};

extern "C" int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

// wf_shade stages a small light BVH in LDS (rt_wavefront.hip, LIGHTS_LDS): 0 when it does not fit RT_SHADE_LIGHTS_F4 pieces
// (or RT_LIGHTS_LDS=0), else 1 + number of inner nodes.
static uint32_t light_lds_inner(const rt::FlatBvh &f) {
    if (const char *e = getenv("RT_LIGHTS_LDS"))
        if (e[0] == '0')
            return 0u;
    if (f.root == RT_NONE || f.tris.empty() || 4u * f.nodes.size() + 4u * f.tris.size() > (size_t)RT_SHADE_LIGHTS_F4)
        return 0u;
    return (uint32_t)f.nodes.size() + 1u;
}

// ---------------------------------------------------------------------------------------------- rt_create, host half
// Everything rt_create derives from the descriptor WITHOUT a device: both reference-topology BVH builds and their flattening
// (or the wide collapse), shading records, materials, the tiled / interleaved texture pool, tables. Built ONCE per rt_create /
// rt_create_on and uploaded to every GPU of a group (a multi-GPU scene used to repeat all of it per GPU: an 8.5 s
// single-thread build times eight on S-10M).
namespace rt {
struct PreparedScene {
    bool dev_build = false, wide_build = false;
    HostBvh host_bvh[2];
    FlatBvh flat[2];            // [0] empty when the scene BVH is built on the device or is wide
    WideBvh wide;               // wide_build && !dev_build
    std::vector<DevTri> wide_tris;
    std::vector<DevAttr> attrs; // scene-BVH order (empty when built on the device)
    std::vector<DevLightAux> laux;
    std::vector<DevMaterial> mats;
    std::vector<DevTexture> texs;
    std::vector<uint32_t> pool;
    std::vector<rt_primitive_desc> prims;
    std::vector<float> lut_lin, lut_gam;
    double build_ms = 0, wide_ms = 0;
    float wide_cost_node = 1.0f, wide_cost_tri = 0.3f;
};
} // namespace rt

// shading records in the order of the triangle records `tris` (to_intersection_info's inputs, bvh.h:80-121)
static void make_attrs(const rt_scene_desc *d, const std::vector<DevTri> &tris, std::vector<DevAttr> &attrs) {
    attrs.resize(tris.size());
    for (size_t k = 0; k < attrs.size(); ++k) {
        const uint32_t t = tris[k].prim;
        DevAttr &a = attrs[k];
        std::memset(&a, 0, sizeof(a));
        std::memcpy(a.n, d->normals + 9 * (size_t)t, 36);
        std::memcpy(a.tg, d->tangents + 9 * (size_t)t, 36);
        std::memcpy(a.uv, d->texcoords + 6 * (size_t)t, 24);
        const DevTri &tr = tris[k];
        V3h c = cross_h({tr.v[0], tr.v[1], tr.v[2]}, {tr.u[0], tr.u[1], tr.u[2]}); // triangle::normal geometry.h:477-479
        float l = len_h(c);
        a.gn[0] = c.x / l;
        a.gn[1] = c.y / l;
        a.gn[2] = c.z / l;
        a.material = d->material_ids[t];
    }
}
// triangle records of the wide tree: the reference's operands a, b - a, c - a (geometry.h:473-475) in the tree's own order
static void make_wide_tris(const rt_scene_desc *d, const rt::WideBvh &wide, std::vector<DevTri> &tris) {
    tris.resize(wide.order.size());
    for (size_t k = 0; k < tris.size(); ++k) {
        const float *p = d->positions + 9 * (size_t)wide.order[k];
        DevTri &t = tris[k];
        for (int c = 0; c < 3; ++c) {
            t.a[c] = p[c];
            t.v[c] = p[3 + c] - p[c];
            t.u[c] = p[6 + c] - p[c];
        }
        t.prim = wide.order[k];
        t.flags = 0;
        t.pad = 0;
    }
}

static int prepare_scene(const rt_scene_desc *d, rt::PreparedScene &P) {
    const uint32_t n = d->n_triangles;
    for (uint32_t i = 0; i < n; ++i)
        if (d->material_ids[i] >= d->n_materials)
            return rt::fail(RT_ERR_INVALID_ARG, "rt_create: material id out of range");
    auto tex_ok = [&](int32_t t) { return t < 0 || (uint32_t)t < d->n_textures; };
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const rt_material_desc &m = d->materials[i];
        if (!tex_ok(m.color_tex) || !tex_ok(m.emissive_tex) || !tex_ok(m.metallic_roughness_tex) || !tex_ok(m.normal_tex))
            return rt::fail(RT_ERR_INVALID_ARG, "rt_create: texture index out of range");
    }
    for (uint32_t i = 0; i < d->n_textures; ++i)
        if (d->textures[i].width == 0 || d->textures[i].height == 0 || !d->textures[i].rgba8)
            return rt::fail(RT_ERR_INVALID_ARG, "rt_create: empty texture");

    // ---- RaytracerStaticContext: scene_bvh over everything, light_bvh over emission != 0 (raytracer.h:441-447)
    std::vector<uint32_t> all(n), lights;
    for (uint32_t i = 0; i < n; ++i) {
        all[i] = i;
        const float *e = d->materials[d->material_ids[i]].emission;
        if (!((e[0] == 0) & (e[1] == 0) & (e[2] == 0)))
            lights.push_back(i);
    }
    const char *env_dev = std::getenv("RT_BVH_DEVICE"), *env_wide = std::getenv("RT_BVH_WIDE");
    P.dev_build = n > 0 && ((d->build_flags & RT_BUILD_DEVICE_LBVH) || (env_dev && std::atoi(env_dev) != 0));
    P.wide_build = n > 0 && ((d->build_flags & RT_BUILD_WIDE) || (env_wide && std::atoi(env_wide) != 0));
    if (const char *e = std::getenv("RT_WIDE_COST_NODE"))
        P.wide_cost_node = (float)std::atof(e);
    if (const char *e = std::getenv("RT_WIDE_COST_TRI"))
        P.wide_cost_tri = (float)std::atof(e);
    // The geometry half — both BVH builds, flattening or the wide collapse, shading records — runs on a thread of its own while
    // this thread lays out the texture pool below (SURVEY 8f-2 "overlap"): on S-sponza the two halves take about as long as
    // each other (BVH 0.10 s, 268 MB of tiled / interleaved texels 0.2 s).
    auto geometry_task = std::async(std::launch::async, [&]() {
        const auto tb0 = std::chrono::steady_clock::now();
        if (!P.dev_build) {
            P.host_bvh[0] = rt::build_bvh(d->positions, n, all);
            if (!P.wide_build) {
                P.flat[0] = rt::flatten_bvh(P.host_bvh[0], d->positions);
                make_attrs(d, P.flat[0].tris, P.attrs);
            }
            P.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count();
            if (P.wide_build) { // production build: collapse the reference-topology tree into the 8-wide quantised tree
                const auto tw0 = std::chrono::steady_clock::now();
                P.wide = rt::build_wide(rt::bin_from_host(P.host_bvh[0]), d->positions, P.wide_cost_node, P.wide_cost_tri);
                make_wide_tris(d, P.wide, P.wide_tris);
                make_attrs(d, P.wide_tris, P.attrs);
                P.wide_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
            }
        }
        P.host_bvh[1] = rt::build_bvh(d->positions, n, lights);
        P.flat[1] = rt::flatten_bvh(P.host_bvh[1], d->positions);
        P.laux.resize(P.flat[1].tris.size());
        for (size_t k = 0; k < P.laux.size(); ++k) {
            const DevTri &tr = P.flat[1].tris[k];
            V3h c = cross_h({tr.v[0], tr.v[1], tr.v[2]}, {tr.u[0], tr.u[1], tr.u[2]});
            float l = len_h(c);
            P.laux[k].normal[0] = c.x / l;
            P.laux[k].normal[1] = c.y / l;
            P.laux[k].normal[2] = c.z / l;
            P.laux[k].area = l / 2; // triangle::square geometry.h:481-483
        }
    });
    struct JoinGeometry { // every return below must wait for the task: it works on `P` and on locals of this frame
        std::future<void> &f;
        ~JoinGeometry() {
            if (f.valid())
                f.wait();
        }
    } join_geometry{geometry_task};

    std::vector<DevMaterial> &mats = P.mats;
    mats.resize(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const rt_material_desc &m = d->materials[i];
        DevMaterial &o = mats[i];
        std::memset(&o, 0, sizeof(o));
        std::memcpy(o.color, m.color, 16);
        std::memcpy(o.emission, m.emission, 12);
        o.roughness = m.roughness;
        o.metallic = m.metallic;
        o.ior = m.ior;
        o.color_tex = m.color_tex;
        o.emissive_tex = m.emissive_tex;
        o.mr_tex = m.metallic_roughness_tex;
        o.normal_tex = m.normal_tex;
    }
    // ---- texture views (rt_device_types.h DevTexture): tiled storage; textures one material samples together at equal
    // size are interleaved record by record. A material slot refers to a VIEW, so materials are remapped here.
    std::vector<DevTexture> &texs = P.texs;
    std::vector<uint32_t> &pool = P.pool;
    {
        auto store = [&](const std::vector<int32_t> &members, uint32_t stride, std::vector<int32_t> &view_of_member) {
            // members: texture ids of equal size (or -1 for an unused slot), one per record dword; returns view ids
            const rt_texture_desc *first = nullptr;
            for (int32_t t : members)
                if (t >= 0)
                    first = &d->textures[t];
            const uint32_t w = first->width, h = first->height;
            const uint32_t tw_log = stride == 1 ? 3u : 2u, th_log = stride == 1 ? 2u : 1u; // 8x4 texels or 4x2 records = 128 B
            const uint32_t tiles_x = (w + (1u << tw_log) - 1) >> tw_log, tiles_y = (h + (1u << th_log) - 1) >> th_log;
            const size_t base = (pool.size() + 31) & ~size_t(31); // 128-B aligned tiles
            const size_t records = (size_t)tiles_x * tiles_y << (tw_log + th_log);
            pool.resize(base + records * stride, 0u);
            view_of_member.assign(members.size(), -1);
            for (size_t k = 0; k < members.size(); ++k) {
                if (members[k] < 0)
                    continue;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(d->textures[members[k]].rgba8);
                for (uint32_t y = 0; y < h; ++y)
                    for (uint32_t x = 0; x < w; ++x) {
                        const size_t tile = (size_t)(y >> th_log) * tiles_x + (x >> tw_log);
                        const size_t within = ((y & ((1u << th_log) - 1)) << tw_log) | (x & ((1u << tw_log) - 1));
                        pool[base + ((tile << (tw_log + th_log)) + within) * stride + k] = src[(size_t)y * w + x];
                    }
                view_of_member[k] = (int32_t)texs.size();
                texs.push_back(DevTexture{w, h, (uint32_t)(base + k), w * h, stride, tiles_x, tw_log, th_log});
            }
        };
        std::vector<int32_t> alone(d->n_textures, -1); // stand-alone view of texture i, built on demand
        auto alone_view = [&](int32_t t) {
            if (alone[t] < 0) {
                std::vector<int32_t> v;
                store({t}, 1, v);
                alone[t] = v[0];
            }
            return alone[t];
        };
        std::vector<std::pair<std::array<int32_t, 4>, std::array<int32_t, 4>>> sets; // slot tuple -> view ids
        const size_t budget = (size_t)1 << 30;                                        // dwords (4 GiB) of interleaved copies
        for (uint32_t i = 0; i < d->n_materials; ++i) {
            DevMaterial &o = mats[i];
            std::array<int32_t, 4> ids = {o.color_tex, o.emissive_tex, o.mr_tex, o.normal_tex}, views = {-1, -1, -1, -1};
            bool found = false;
            for (const auto &e : sets)
                if (e.first == ids) {
                    views = e.second;
                    found = true;
                }
            if (!found) {
                // the largest group of this material's textures that share a size (> 1x1) is interleaved; the rest stand alone
                std::array<int32_t, 4> group = {-1, -1, -1, -1};
                int best_n = 0;
                for (int a = 0; a < 4; ++a) {
                    if (ids[a] < 0 || d->textures[ids[a]].width * d->textures[ids[a]].height == 1)
                        continue;
                    std::array<int32_t, 4> g = {-1, -1, -1, -1};
                    int n = 0;
                    for (int b = 0; b < 4; ++b)
                        if (ids[b] >= 0 && d->textures[ids[b]].width == d->textures[ids[a]].width && d->textures[ids[b]].height == d->textures[ids[a]].height) {
                            g[b] = ids[b];
                            ++n;
                        }
                    if (n > best_n) {
                        best_n = n;
                        group = g;
                    }
                }
                if (best_n >= 2 && pool.size() < budget) {
                    std::vector<int32_t> v;
                    store({group[0], group[1], group[2], group[3]}, 4, v);
                    for (int a = 0; a < 4; ++a)
                        views[a] = v[a];
                }
                for (int a = 0; a < 4; ++a)
                    if (ids[a] >= 0 && views[a] < 0)
                        views[a] = alone_view(ids[a]);
                sets.push_back({ids, views});
            }
            o.color_tex = views[0];
            o.emissive_tex = views[1];
            o.mr_tex = views[2];
            o.normal_tex = views[3];
        }
        if (pool.size() >= ((size_t)1 << 32))
            return rt::fail(RT_ERR_OOM, "rt_create: texture pool exceeds 2^32 texels");
    }
    // ---- analytic primitives of the scene-txt front end (rt_primspec.h)
    if (d->n_primitives) {
        if (!d->primitives || d->n_primitives > RT_MAX_PRIMITIVES)
            return rt::fail(RT_ERR_INVALID_ARG, "rt_create: bad primitive list (limit " + std::to_string(RT_MAX_PRIMITIVES) + ")");
        P.prims.assign(d->primitives, d->primitives + d->n_primitives);
        for (const rt_primitive_desc &pr : P.prims)
            if ((pr.kind != RT_PRIM_ELLIPSOID && pr.kind != RT_PRIM_PLANE) || pr.material_id >= d->n_materials)
                return rt::fail(RT_ERR_INVALID_ARG, "rt_create: primitive with unknown kind or material id out of range");
    }
    P.lut_lin.resize(256);
    P.lut_gam.resize(256);
    for (int k = 0; k < 256; ++k) {
        P.lut_lin[k] = k / 255.0f;                 // Texture::load_img geometry.h:593-594
        P.lut_gam[k] = std::pow(P.lut_lin[k], 2.2f); // rgba_apply_gamma geometry.h:525-527 (float powf)
    }
    geometry_task.get();
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------- rt_create, device half
static int create_impl(const rt_scene_desc *d, const std::shared_ptr<const rt::PreparedScene> &prep, int device, rt_scene *s) {
    const rt::PreparedScene &P = *prep;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return rt::fail(RT_ERR_NO_DEVICE, "rt_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_create: device ordinal out of range");
    s->device = device;
    s->prep = prep;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    s->num_cus = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&s->ev0));
    HIP_TRY(hipEventCreate(&s->ev1));

    const uint32_t n = d->n_triangles;
    const bool dev_build = P.dev_build, wide_build = P.wide_build;
    rt::DeviceBvh dbvh{};
    // a wide tree collapsed from the DEVICE-built binary tree belongs to this GPU's build (the host-built one is in `P`)
    rt::WideBvh wide_local;
    std::vector<DevTri> wide_tris_local;
    std::vector<DevAttr> attrs_local;
    const rt::WideBvh *wide = &P.wide;
    const std::vector<DevTri> *wide_tris = &P.wide_tris;
    const std::vector<DevAttr> *attrs = &P.attrs;
    s->build_ms = P.build_ms;
    s->wide_ms = P.wide_ms;
    if (dev_build) {
        // ---- the scene BVH, the triangle records and the shading records are built on the device (rt_bvh_device.hip)
        const char *what = "";
        const char *env_hc = std::getenv("RT_WIDE_HOST_COLLAPSE"); // development: collapse the device tree on the host instead
        const bool device_collapse = wide_build && !(env_hc && std::atoi(env_hc) != 0);
        hipError_t be = rt::build_bvh_device(d, s->stream, &dbvh, &what, device_collapse, P.wide_cost_node, P.wide_cost_tri);
        if (be != hipSuccess)
            return rt::fail(be == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, std::string("device BVH build: ") + what + ": " + hipGetErrorString(be));
        s->device_built = true;
        s->build_ms = dbvh.build_ms;
        s->build_upload_ms = dbvh.upload_ms;
        if (!wide_build) {
            s->owned.push_back(dbvh.nodes);
            s->owned.push_back(dbvh.tris);
            s->owned.push_back(dbvh.attrs);
        } else if (dbvh.wide) {
            // the whole production build ran on the device (LBVH, dynamic program in the refit, level-by-level emission)
            s->owned.push_back(dbvh.wide);
            s->owned.push_back(dbvh.tris);
            s->owned.push_back(dbvh.attrs);
            s->wide_ms = dbvh.wide_ms;
        } else {
            // production build on top of the device tree: read it back, collapse on the host, upload the wide tree
            const auto tw0 = std::chrono::steady_clock::now();
            std::vector<DevNode> hn(dbvh.n_inner);
            std::vector<DevTri> ht(dbvh.n_tris);
            hipError_t ce = hipSuccess;
            if (dbvh.n_inner)
                ce = hipMemcpy(hn.data(), dbvh.nodes, sizeof(DevNode) * hn.size(), hipMemcpyDeviceToHost);
            if (ce == hipSuccess && dbvh.n_tris)
                ce = hipMemcpy(ht.data(), dbvh.tris, sizeof(DevTri) * ht.size(), hipMemcpyDeviceToHost);
            (void)hipFree(dbvh.nodes);
            (void)hipFree(dbvh.tris);
            (void)hipFree(dbvh.attrs);
            dbvh.nodes = nullptr, dbvh.tris = nullptr, dbvh.attrs = nullptr;
            if (ce != hipSuccess)
                return rt::fail(RT_ERR_HIP, std::string("wide build: reading the device BVH back: ") + hipGetErrorString(ce));
            wide_local = rt::build_wide(rt::bin_from_device(hn, ht, dbvh.root), d->positions, P.wide_cost_node, P.wide_cost_tri);
            make_wide_tris(d, wide_local, wide_tris_local);
            make_attrs(d, wide_tris_local, attrs_local);
            wide = &wide_local, wide_tris = &wide_tris_local, attrs = &attrs_local;
            s->wide_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
        }
    }
    const bool wide_on_device = wide_build && dbvh.wide != nullptr;
    if (wide_build) {
        s->wide_built = true;
        s->wide_depth = wide_on_device ? dbvh.wide_depth : wide->depth;
        s->wide_cost = wide_on_device ? 0.0 : wide->sah_cost;
    }
    DevScene &D = s->dev;
    int rc;
    for (int w = 0; w < 2; ++w) {
        DevBvh &b = w == 0 ? D.scene : D.lights;
        if (w == 0 && wide_on_device) {
            b.wide = dbvh.wide;
            b.tris = dbvh.tris;
            b.nodes = nullptr;
            b.root = 0u;
            b.n_tris = dbvh.n_tris;
            b.n_wide = dbvh.n_wide;
            b.fast_ok = 0u;
            s->dev_n_inner[0] = 0;
            continue;
        }
        if (w == 0 && wide_build) {
            if ((rc = upload(wide->nodes, &b.wide, s->owned)) != RT_OK)
                return rc;
            if ((rc = upload(*wide_tris, &b.tris, s->owned)) != RT_OK)
                return rc;
            b.nodes = nullptr;
            b.root = wide->nodes.empty() ? RT_NONE : 0u;
            b.n_tris = (uint32_t)wide_tris->size();
            b.n_wide = (uint32_t)wide->nodes.size();
            b.fast_ok = 0u;
            s->dev_n_inner[0] = 0;
            continue;
        }
        if (w == 0 && dev_build) {
            b.nodes = dbvh.nodes;
            b.tris = dbvh.tris;
            b.root = dbvh.root;
            b.n_tris = dbvh.n_tris;
            b.fast_ok = dbvh.fast_ok ? 1u : 0u;
            s->dev_n_inner[0] = dbvh.n_inner;
            continue;
        }
        if ((rc = upload(P.flat[w].nodes, &b.nodes, s->owned)) != RT_OK)
            return rc;
        if ((rc = upload(P.flat[w].tris, &b.tris, s->owned)) != RT_OK)
            return rc;
        b.root = P.flat[w].root;
        b.n_tris = (uint32_t)P.flat[w].tris.size();
        b.fast_ok = P.flat[w].fast_ok ? 1u : 0u;
        b.lds_inner = w == 1 ? light_lds_inner(P.flat[w]) : 0u;
        s->dev_n_inner[w] = (uint32_t)P.flat[w].nodes.size();
    }
    if (dev_build && (!wide_build || wide_on_device))
        D.attrs = dbvh.attrs;
    else if ((rc = upload(*attrs, &D.attrs, s->owned)) != RT_OK)
        return rc;
    if ((rc = upload(P.laux, &D.light_aux, s->owned)) != RT_OK)
        return rc;
    if ((rc = upload(P.mats, &D.materials, s->owned)) != RT_OK)
        return rc;
    if ((rc = upload(P.texs, &D.textures, s->owned)) != RT_OK)
        return rc;
    if ((rc = upload(P.pool, &D.texels, s->owned)) != RT_OK)
        return rc;
    if ((rc = upload(P.lut_lin, &D.lut_linear, s->owned)) != RT_OK)
        return rc;
    if ((rc = upload(P.lut_gam, &D.lut_gamma, s->owned)) != RT_OK)
        return rc;
    if ((rc = upload(P.prims, &D.prims, s->owned)) != RT_OK)
        return rc;
    D.n_prims = (uint32_t)P.prims.size();
    D.n_triangles = n;
    std::memcpy(D.cam_pos, d->camera.position, 12);
    std::memcpy(D.cam_right, d->camera.right, 12);
    std::memcpy(D.cam_up, d->camera.up, 12);
    std::memcpy(D.cam_fwd, d->camera.forward, 12);
    std::memcpy(D.bg, d->bg_color, 12);
    D.ray_depth = d->ray_depth;
    for (int k = 0; k < 3; ++k) { // scene bounds for the ray-ordering key (root node box of the scene BVH)
        float lo = 0.f, hi = 1.f;
        if (dev_build) {
            lo = dbvh.lo[k];
            hi = dbvh.hi[k];
        } else if (P.host_bvh[0].root != RT_NONE && !P.host_bvh[0].nodes.empty()) {
            lo = P.host_bvh[0].nodes[P.host_bvh[0].root].lo[k];
            hi = P.host_bvh[0].nodes[P.host_bvh[0].root].hi[k];
        }
        D.bounds_lo[k] = lo;
        D.bounds_inv[k] = (hi > lo) ? 1.0f / (hi - lo) : 0.0f;
    }
    s->cam = d->camera;

    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, sizeof(uint32_t) * 64));
    s->owned.push_back(p);
    s->d_counter = static_cast<uint32_t *>(p);
    HIP_TRY(hipMalloc(&p, sizeof(DevStats)));
    s->owned.push_back(p);
    s->d_stats = static_cast<DevStats *>(p);
    // the uploads above went through the null stream; the scene's stream is non-blocking (not ordered with it), so make
    // sure everything has landed before the first kernel can be launched
    HIP_TRY(hipDeviceSynchronize());
    return RT_OK;
}

static int check_desc(const rt_scene_desc *desc, const void *out) {
    if (!desc || !out)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_create: null argument");
    if (desc->abi_version != RT_ABI_VERSION)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_create: ABI version mismatch");
    if (desc->ray_depth > RT_MAX_RAY_DEPTH)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_create: ray_depth above RT_MAX_RAY_DEPTH (32)");
    if (desc->n_triangles && (!desc->positions || !desc->normals || !desc->texcoords || !desc->tangents || !desc->material_ids || !desc->materials))
        return rt::fail(RT_ERR_INVALID_ARG, "rt_create: null geometry array");
    if (desc->n_triangles > RT_LEAF_BEGIN_MASK)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_create: too many triangles (limit 2^27 - 1)");
    return RT_OK;
}

// one replica of a prepared scene on one device (rt_group.cpp builds a multi-GPU scene out of these)
int rt::create_replica(const rt_scene_desc *desc, const std::shared_ptr<const rt::PreparedScene> &prep, int device, rt_scene **out) {
    rt_scene *s = new rt_scene();
    int rc = create_impl(desc, prep, device, s);
    if (rc != RT_OK) {
        delete s;
        return rc;
    }
    *out = s;
    return RT_OK;
}
int rt::prepare(const rt_scene_desc *desc, std::shared_ptr<const rt::PreparedScene> *out) {
    auto P = std::make_shared<rt::PreparedScene>();
    if (int rc = prepare_scene(desc, *P); rc != RT_OK)
        return rc;
    *out = P;
    return RT_OK;
}

extern "C" int rt_create_on(const rt_scene_desc *desc, const int *devices, int n_devices, rt_scene **out) {
    if (int rc = check_desc(desc, out); rc != RT_OK)
        return rc;
    if (!devices || n_devices < 1)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_create_on: empty device list");
    rt::Group *g = nullptr;
    if (int rc = rt::group_create(desc, devices, n_devices, &g); rc != RT_OK)
        return rc;
    rt_scene *s = new rt_scene();
    s->group = g;
    s->device = devices[0];
    *out = s;
    return RT_OK;
}

extern "C" int rt_create(const rt_scene_desc *desc, int device, rt_scene **out) {
    if (int rc = check_desc(desc, out); rc != RT_OK)
        return rc;
    if (device == RT_ALL_DEVICES) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            return rt::fail(RT_ERR_NO_DEVICE, "rt_create: no HIP device available (this library has no CPU fallback)");
        std::vector<int> all(ndev);
        for (int i = 0; i < ndev; ++i)
            all[i] = i;
        return rt_create_on(desc, all.data(), ndev, out);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) // before any host work: without a GPU nothing below can succeed
        return rt::fail(RT_ERR_NO_DEVICE, "rt_create: no HIP device available (this library has no CPU fallback)");
    std::shared_ptr<const rt::PreparedScene> prep;
    if (int rc = rt::prepare(desc, &prep); rc != RT_OK)
        return rc;
    return rt::create_replica(desc, prep, device, out);
}

extern "C" int rt_scene_device_count(const rt_scene *scene) { return !scene ? 0 : (scene->group ? rt::group_size(scene->group) : 1); }

extern "C" void rt_destroy(rt_scene *scene) { delete scene; }

// rt_render (fb_rgb: linear float3) and rt_render_rgb8 (rgb8_out: the tone-mapped image, film on the device)
static int render_impl(rt_scene *s, const rt_params *p, float *fb_rgb, uint8_t *rgb8_out, rt_stats *stats) {
    if (!s || !p || (!fb_rgb && !rgb8_out))
        return rt::fail(RT_ERR_INVALID_ARG, "rt_render: null argument");
    if (s->group)
        return rt::group_render(s->group, p, fb_rgb, rgb8_out, stats);
    if (p->width == 0 || p->height == 0 || (uint64_t)p->width * p->height >= 0x7FFFFFFFull)
        return rt::fail(RT_ERR_INVALID_ARG, "Illegal image size" + std::to_string(p->width) + "x" + std::to_string(p->height)); // image.h:26
    if (p->rng_mode != RT_RNG_DEVICE && p->rng_mode != RT_RNG_REFERENCE)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_render: unknown rng_mode");
    if (p->samples == 0)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_render: samples must be >= 1");
    auto wall0 = std::chrono::steady_clock::now();
    if (stats)
        std::memset(stats, 0, sizeof(*stats));
    if (s->dev.ray_depth == 0) // raytracer.h:630-631
        return RT_OK;
    HIP_TRY(hipSetDevice(s->device));

    const uint32_t n_pix = p->width * p->height;
    RenderLaunch L{};
    L.width = p->width;
    L.height = p->height;
    L.samples = p->samples;
    L.rng_mode = p->rng_mode;
    L.seed = p->seed;
    L.shard_count = p->shard_count ? p->shard_count : 1;
    L.shard_index = p->shard_index;
    if (L.shard_index >= L.shard_count)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_render: shard_index >= shard_count");
    const uint32_t unit = p->rng_mode == RT_RNG_REFERENCE ? RT_SPAN : 1u;
    uint32_t block = p->shard_block ? p->shard_block : (L.shard_count > 1 ? RT_SPAN * 8u : n_pix);
    if (L.shard_count == 1)
        block = ((n_pix + unit - 1) / unit) * unit; // one block holding everything
    if (block % unit != 0)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_render: shard_block must be a multiple of 256 in reference RNG mode");
    L.shard_block = block;
    L.items_per_block = block / unit;
    // work items of this shard: blocks b = shard_index, shard_index + count, ... below ceil(n_pix / block)
    const uint64_t n_blocks = ((uint64_t)n_pix + block - 1) / block;
    uint64_t items = 0, local_pixels = 0;
    for (uint64_t b = L.shard_index; b < n_blocks; b += L.shard_count) {
        uint64_t first = b * block, last = std::min<uint64_t>(first + block, n_pix);
        items += (last - first + unit - 1) / unit;
        local_pixels += last - first;
    }
    L.n_items = (uint32_t)items;
    // gen_ray's tan terms (raytracer.h:531-535) and Camera::fov_y (scene.h:69-71), float overloads
    L.tan_x = std::tan(s->cam.fov_x / 2);
    const float fov_y = std::atan(std::tan(s->cam.fov_x / 2) * p->height / p->width) * 2;
    L.tan_y = std::tan(fov_y / 2);

    const bool device_fb = (p->flags & RT_FLAG_DEVICE_FB) != 0;
    const size_t fb_floats = (size_t)n_pix * 3;
    float *d_fb = fb_rgb;
    if (!device_fb || rgb8_out) { // the float framebuffer is internal unless the caller keeps it in HBM
        int rc = s->ensure_fb(fb_floats);
        if (rc != RT_OK)
            return rc;
        d_fb = s->d_fb;
    }
    uint8_t *d_rgb8 = nullptr;
    if (rgb8_out) {
        int rc = s->ensure_film(device_fb ? 0 : fb_floats);
        if (rc != RT_OK)
            return rc;
        d_rgb8 = device_fb ? rgb8_out : s->d_rgb8;
    }
    L.fb = d_fb;
    L.counter = s->d_counter;
    const bool counters = stats && (p->flags & RT_FLAG_COUNTERS);
    L.stats = counters ? s->d_stats : nullptr;
    HIP_TRY(hipMemsetAsync(s->d_counter, 0, sizeof(uint32_t), s->stream));
    if (counters)
        HIP_TRY(hipMemsetAsync(s->d_stats, 0, sizeof(DevStats), s->stream));

    int blocks = (int)std::min<uint64_t>((items + 255) / 256, (uint64_t)s->num_cus * s->blocks_per_cu);
    if (blocks < 1)
        blocks = 1;
    const bool wavefront = p->rng_mode == RT_RNG_DEVICE && !(p->flags & RT_FLAG_MEGAKERNEL);
    if (s->wide_built && !wavefront)
        return rt::fail(RT_ERR_UNSUPPORTED, "rt_render: a scene built with RT_BUILD_WIDE renders through the wavefront pipeline only (the megakernel and "
                                            "the reference-RNG parity mode walk the reference's binary tree: create the scene without RT_BUILD_WIDE)");
    s->ext_events.reset();
    // everything from here to the final synchronisation is queued on the scene's stream; a failure in between must not
    // return while kernels are still in flight (a later ensure_wavefront / rt_destroy would free memory under them)
    auto queue_render = [&]() -> int {
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
    if (L.n_items > 0 && wavefront) {
        // ---- production path: wavefront pipeline over (pixel tile) x (sample range) passes, all stream-ordered
        // Paths per pass: the larger a pass, the smaller the share of each bounce launch's drain phase (measured on
        // S-sponza 1000x1000x64: 8 M paths 148, 16 M 158, 32 M 164, 64 M 167 Msamples/s). 64 M paths x 432 B = 29 GB
        // of workspace, sized for 288 GB of HBM; capped by free device memory below.
        uint64_t max_paths = 64ull << 20;
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const uint64_t per_path = 2 * sizeof(WfPath) + sizeof(WfHit) + s->dev.ray_depth * sizeof(WfFold) + sizeof(RtF4) + 16 /* sort keys + slots */;
                const uint64_t have = free_b + s->wf_paths_cap * per_path; // what is already ours can be reused
                const uint64_t fit = (uint64_t)(0.6 * (double)have) / per_path;
                max_paths = std::max<uint64_t>(1u << 16, std::min(max_paths, fit));
            }
        }
        if (const char *e = std::getenv("RT_WF_MAX_PATHS"))
            max_paths = std::max<uint64_t>(1024, std::strtoull(e, nullptr, 0));
        const uint64_t tile_pixels = std::min<uint64_t>(local_pixels, max_paths);
        const uint32_t pass_spp = (uint32_t)std::clamp<uint64_t>(max_paths / tile_pixels, 1, p->samples);
        int rc = s->ensure_wavefront(tile_pixels * pass_spp, tile_pixels, s->dev.ray_depth);
        if (rc != RT_OK)
            return rc;
        WfLaunch W{};
        W.width = p->width;
        W.height = p->height;
        W.samples = p->samples;
        W.shard_index = L.shard_index;
        W.shard_count = L.shard_count;
        W.shard_block = L.shard_block;
        W.ray_depth = s->dev.ray_depth;
        W.seed = p->seed;
        W.tan_x = L.tan_x;
        W.tan_y = L.tan_y;
        s->wf_bind(W);
        W.fb = d_fb;
        const char *sort_env = std::getenv("RT_WF_SORT");
        if (!s->wf_host_count && hipHostMalloc((void **)&s->wf_host_count, 48 * sizeof(uint32_t)) != hipSuccess)
            s->wf_host_count = nullptr;
        while (s->wf_count_events.size() < RT_MAX_RAY_DEPTH + 1) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
                break;
            s->wf_count_events.push_back(ev);
        }
        rt::WfHostSync hsync{s->wf_host_count, s->wf_count_events.data(), (int)s->wf_count_events.size()};
        // Coherence sort of bounces >= 1 (24-bit key: cell, octant, direction sub-cone: measured best). It pays where node fetches
        // miss the caches; a WIDE tree that fits them gains nothing from it and loses the sort's own time (S-sponza, 15 MB of
        // nodes + triangles: 436 vs 409 Msamples/s without / with; S-10M, 0.7 GB: 196 vs 220; profiles/r03_wide.txt).
        const size_t bvh_bytes = (size_t)s->dev.scene.n_wide * sizeof(WideNode) + (size_t)s->dev.scene.n_tris * sizeof(DevTri);
        const uint32_t sort_default = (s->wide_built && bvh_bytes < ((size_t)128 << 20)) ? 0u : 4u;
        W.sort_mode = sort_env ? (uint32_t)std::atoi(sort_env) : sort_default;
        // production traversal: global-best pruning (rt_abi.h RT_FLAG_GLOBAL_BEST; RT_TRAVERSAL=global for callers without flags)
        const char *trav_env = std::getenv("RT_TRAVERSAL");
        W.global_best = ((p->flags & RT_FLAG_GLOBAL_BEST) || (trav_env && !std::strcmp(trav_env, "global"))) ? 1u : 0u;
        // primary rays as packets (wf_extend_packet): RT_WF_PACKET=0 never, =1 always; default: from 16 samples per pixel and pass
        // up, until the kernel's census says its packets fall apart (fewer than 33 of 64 lanes served per trip: the measured
        // break-even against wf_extend, profiles/r02_packet.txt) for this image size / samples per pass
        const char *pkt_env = std::getenv("RT_WF_PACKET");
        const int pkt_mode = pkt_env ? std::atoi(pkt_env) : -1;
        W.stats = L.stats;
        for (uint64_t p0 = 0; p0 < local_pixels; p0 += tile_pixels) {
            W.first_pixel = (uint32_t)p0;
            W.pass_pixels = (uint32_t)std::min<uint64_t>(tile_pixels, local_pixels - p0);
            for (uint32_t s0 = 0; s0 < p->samples; s0 += pass_spp) {
                W.first_sample = s0;
                W.pass_samples = std::min<uint32_t>(pass_spp, p->samples - s0);
                W.n_paths = W.pass_pixels * W.pass_samples;
                const uint64_t key = ((uint64_t)p->width << 44) ^ ((uint64_t)p->height << 24) ^ ((uint64_t)W.pass_samples << 8) ^ (uint64_t)L.shard_count;
                if (key != s->pkt_key) {
                    s->pkt_key = key;
                    s->pkt_off = false;
                }
                const uint32_t pkt_min_spp = s->wide_built ? 4u : 16u; // measured break-even of the two packet kernels
                W.use_packet = pkt_mode == 0 ? 0u : pkt_mode > 0 ? 1u : (W.pass_samples >= pkt_min_spp && !s->pkt_off) ? 1u : 0u;
                unsigned long long census[2] = {0ull, 0ull};
                HIP_TRY(rt::launch_wavefront_pass(s->dev, W, counters, s->num_cus, s0 == 0, s0 + W.pass_samples >= p->samples, s->stream,
                                                  stats ? &s->ext_events : nullptr, census, s->wf_host_count ? &hsync : nullptr));
                // lanes served per packet trip below which the per-lane kernel is faster: 33 for wf_extend_packet (profiles/r02_packet.txt);
                // RT_WF_PACKET_MIN overrides it (development). The wide packet kernel's break-even is lower (profiles/r03_wide.txt).
                static const double min_lanes_env = std::getenv("RT_WF_PACKET_MIN") ? std::atof(std::getenv("RT_WF_PACKET_MIN")) : -1.0;
                const double min_lanes = min_lanes_env >= 0.0 ? min_lanes_env : (s->wide_built ? RT_WIDE_PACKET_MIN_LANES : 33.0);
                if (census[0] != 0ull) {
                    s->pkt_lanes_x100 = (uint32_t)(100.0 * (double)census[1] / (double)census[0]);
                    if ((double)census[1] < min_lanes * (double)census[0])
                        s->pkt_off = true;
                }
            }
        }
    } else if (L.n_items > 0) {
        // ---- persistent megakernel: reference-RNG parity mode, or RT_FLAG_MEGAKERNEL cross-check
        HIP_TRY(rt::launch_render(s->dev, L, counters, blocks, s->stream));
    }
    if (rgb8_out && L.n_items > 0) // film on the device: this shard's pixels -> rgb8 (image.h:49-82)
        HIP_TRY(rt::launch_film(d_fb, d_rgb8, n_pix, L.shard_index, L.shard_count, block, s->d_film_table, s->stream));
    HIP_TRY(hipEventRecord(s->ev1, s->stream));
    return RT_OK;
    };
    if (int rc = queue_render(); rc != RT_OK) {
        (void)hipStreamSynchronize(s->stream);
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));

    if (!device_fb && L.n_items > 0) { // copy back only this shard's blocks; other pixels of the destination stay untouched
        const size_t elem = rgb8_out ? 1 : sizeof(float);
        char *dst = rgb8_out ? reinterpret_cast<char *>(rgb8_out) : reinterpret_cast<char *>(fb_rgb);
        const char *src = rgb8_out ? reinterpret_cast<const char *>(d_rgb8) : reinterpret_cast<const char *>(d_fb);
        if (L.shard_count == 1) {
            HIP_TRY(hipMemcpy(dst, src, fb_floats * elem, hipMemcpyDeviceToHost));
        } else {
            for (uint64_t b = L.shard_index; b < n_blocks; b += L.shard_count) {
                uint64_t first = b * block, last = std::min<uint64_t>(first + block, n_pix);
                HIP_TRY(hipMemcpy(dst + 3 * first * elem, src + 3 * first * elem, (last - first) * 3 * elem, hipMemcpyDeviceToHost));
            }
        }
    }
    if (stats) {
        DevStats h{};
        if (counters)
            HIP_TRY(hipMemcpy(&h, s->d_stats, sizeof(h), hipMemcpyDeviceToHost));
        stats->samples = counters ? h.samples : local_pixels * p->samples;
        stats->casts = h.casts;
        stats->nodes_visited = h.nodes;
        stats->box_tests = h.box_tests;
        stats->tri_tests = h.tri_tests;
        stats->shaded_hits = h.shaded;
        stats->light_queries = h.lq;
        stats->light_nodes = h.lnodes;
        stats->light_box_tests = h.lbox;
        stats->light_tri_tests = h.ltri;
        stats->light_hits = h.lhits;
        stats->texel_fetches = h.texels;
        stats->kernel_ms = ms;
        const std::vector<hipEvent_t> &xe = s->ext_events.ev;
        const size_t n_xe = s->ext_events.used & ~size_t(1);
        stats->dominant_launches = wavefront ? (uint32_t)(n_xe / 2) : (L.n_items > 0 ? 1u : 0u);
        stats->dominant_ms = wavefront ? 0.0 : ms;
        for (size_t i = 0; i + 1 < n_xe; i += 2) {
            float t = 0;
            if (hipEventElapsedTime(&t, xe[i], xe[i + 1]) == hipSuccess)
                stats->dominant_ms += t;
        }
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        stats->reserved = wavefront ? s->pkt_lanes_x100 : 0u; // development: the packet kernel's census (lanes served per trip x 100; 0 = it did not run)
    }
    return RT_OK;
}

extern "C" int rt_render(rt_scene *s, const rt_params *p, float *fb_rgb, rt_stats *stats) {
    if (!fb_rgb)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_render: null argument");
    return render_impl(s, p, fb_rgb, nullptr, stats);
}

extern "C" int rt_render_rgb8(rt_scene *s, const rt_params *p, uint8_t *rgb8, rt_stats *stats) {
    if (!rgb8)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_render_rgb8: null argument");
    return render_impl(s, p, nullptr, rgb8, stats);
}

extern "C" int rt_film_rgb8(rt_scene *s, const float *rgb, size_t n_pixels, uint8_t *out_rgb8) {
    if (s && s->group)
        return rt_film_rgb8(rt::group_primary(s->group), rgb, n_pixels, out_rgb8);
    if (!s || (n_pixels && (!rgb || !out_rgb8)) || n_pixels >= 0x7FFFFFFFull)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_film_rgb8: bad argument");
    if (n_pixels == 0)
        return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    int rc = s->ensure_fb(n_pixels * 3);
    if (rc == RT_OK)
        rc = s->ensure_film(n_pixels * 3);
    if (rc != RT_OK)
        return rc;
    HIP_TRY(hipMemcpyAsync(s->d_fb, rgb, n_pixels * 3 * sizeof(float), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(rt::launch_film(s->d_fb, s->d_rgb8, (uint32_t)n_pixels, 0, 1, (uint32_t)n_pixels, s->d_film_table, s->stream));
    HIP_TRY(hipMemcpyAsync(out_rgb8, s->d_rgb8, n_pixels * 3, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return RT_OK;
}

extern "C" int rt_cast_rays(rt_scene *s, const float *rays, uint32_t n, uint32_t *prim_out, float *bct_out) {
    if (s && s->group)
        return rt_cast_rays(rt::group_primary(s->group), rays, n, prim_out, bct_out);
    if (!s || (n && (!rays || !prim_out || !bct_out)))
        return rt::fail(RT_ERR_INVALID_ARG, "rt_cast_rays: null argument");
    if (n == 0)
        return RT_OK;
    if (s->wide_built) // no binary tree in HBM: the probe goes through the renderer's own wide kernel
        return rt_cast_rays_ex(s, rays, n, RT_CAST_EXTEND, prim_out, bct_out, nullptr);
    HIP_TRY(hipSetDevice(s->device));
    DevBuf b_rays, b_bct, b_prim;
    HIP_TRY(b_rays.alloc((size_t)n * 24));
    HIP_TRY(b_bct.alloc((size_t)n * 12));
    HIP_TRY(b_prim.alloc((size_t)n * 4));
    float *d_rays = b_rays.as<float>(), *d_bct = b_bct.as<float>();
    uint32_t *d_prim = b_prim.as<uint32_t>();
    int rc = RT_OK;
    // every copy is ordered on the scene's own (non-blocking) stream with the kernel
    hipError_t e = hipMemcpyAsync(d_rays, rays, (size_t)n * 24, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess)
        e = rt::launch_cast(s->dev, d_rays, n, d_prim, d_bct, s->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(prim_out, d_prim, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(bct_out, d_bct, (size_t)n * 12, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess)
        rc = rt::fail(RT_ERR_HIP, std::string("rt_cast_rays: ") + hipGetErrorString(e));
    return rc;
}

extern "C" int rt_cast_rays_ex(rt_scene *s, const float *rays, uint32_t n, uint32_t mode, uint32_t *prim_out, float *bct_out, rt_stats *stats) {
    if (s && s->group)
        return rt_cast_rays_ex(rt::group_primary(s->group), rays, n, mode, prim_out, bct_out, stats);
    if (mode == RT_CAST_PROBE && !(s && s->wide_built)) {
        if (stats)
            std::memset(stats, 0, sizeof(*stats));
        return rt_cast_rays(s, rays, n, prim_out, bct_out);
    }
    if (!s || (n && (!rays || !prim_out || !bct_out)) || mode > RT_CAST_PACKET_GLOBAL)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_cast_rays_ex: bad argument");
    if (stats)
        std::memset(stats, 0, sizeof(*stats));
    if (n == 0)
        return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    if (int rc = s->ensure_wavefront(n, 1, 1); rc != RT_OK)
        return rc;
    DevBuf b_rays, b_bct, b_prim;
    HIP_TRY(b_rays.alloc((size_t)n * 24));
    HIP_TRY(b_bct.alloc((size_t)n * 12));
    HIP_TRY(b_prim.alloc((size_t)n * 4));
    WfLaunch W{};
    s->wf_bind(W);
    W.ray_depth = 1;
    W.global_best = (mode == RT_CAST_EXTEND_GLOBAL || mode == RT_CAST_PACKET_GLOBAL) ? 1u : 0u;
    W.stats = stats ? s->d_stats : nullptr;
    const bool packet = mode == RT_CAST_PACKET || mode == RT_CAST_PACKET_GLOBAL;
    hipError_t e = hipMemcpyAsync(b_rays.p, rays, (size_t)n * 24, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess && stats)
        e = hipMemsetAsync(s->d_stats, 0, sizeof(DevStats), s->stream);
    if (e == hipSuccess)
        e = hipEventRecord(s->ev0, s->stream);
    if (e == hipSuccess)
        e = rt::launch_wavefront_cast(s->dev, W, b_rays.as<float>(), n, packet, stats != nullptr, b_prim.as<uint32_t>(), b_bct.as<float>(), s->stream);
    if (e == hipSuccess)
        e = hipEventRecord(s->ev1, s->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(prim_out, b_prim.p, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(bct_out, b_bct.p, (size_t)n * 12, hipMemcpyDeviceToHost, s->stream);
    DevStats h{};
    if (e == hipSuccess && stats)
        e = hipMemcpyAsync(&h, s->d_stats, sizeof(h), hipMemcpyDeviceToHost, s->stream);
    const hipError_t se = hipStreamSynchronize(s->stream); // also on failure: nothing may stay in flight over the DevBufs
    if (e == hipSuccess)
        e = se;
    if (e != hipSuccess)
        return rt::fail(RT_ERR_HIP, std::string("rt_cast_rays_ex: ") + hipGetErrorString(e));
    if (stats) {
        stats->casts = n;
        stats->nodes_visited = h.nodes;
        stats->box_tests = h.box_tests;
        stats->tri_tests = h.tri_tests;
        stats->light_queries = h.lq, stats->light_nodes = h.lnodes, stats->light_box_tests = h.lbox, stats->light_tri_tests = h.ltri, stats->light_hits = h.lhits; // 0 unless a census build
        float ms = 0;
        if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess)
            stats->kernel_ms = ms;
    }
    return RT_OK;
}

extern "C" int rt_light_pdf(rt_scene *s, const float *rays, uint32_t n, float *pdf_out) {
    if (s && s->group)
        return rt_light_pdf(rt::group_primary(s->group), rays, n, pdf_out);
    if (!s || (n && (!rays || !pdf_out)))
        return rt::fail(RT_ERR_INVALID_ARG, "rt_light_pdf: null argument");
    if (n == 0)
        return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    DevBuf b_rays, b_pdf;
    HIP_TRY(b_rays.alloc((size_t)n * 24));
    HIP_TRY(b_pdf.alloc((size_t)n * 4));
    float *d_rays = b_rays.as<float>(), *d_pdf = b_pdf.as<float>();
    int rc = RT_OK;
    hipError_t e = hipMemcpyAsync(d_rays, rays, (size_t)n * 24, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess)
        e = rt::launch_light_pdf(s->dev, d_rays, n, d_pdf, s->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(pdf_out, d_pdf, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess)
        rc = rt::fail(RT_ERR_HIP, std::string("rt_light_pdf: ") + hipGetErrorString(e));
    return rc;
}

extern "C" int rt_bvh_device_dump(rt_scene *s, int which, uint32_t *n_inner, uint32_t *n_tris, uint32_t *root, uint32_t *nodes64, uint32_t *tris48) {
    if (!s || which < 0 || which > 1)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_bvh_device_dump: bad argument");
    if (s->group)
        return rt_bvh_device_dump(rt::group_primary(s->group), which, n_inner, n_tris, root, nodes64, tris48);
    if (which == 0 && s->wide_built)
        return rt::fail(RT_ERR_UNSUPPORTED, "rt_bvh_device_dump: the scene BVH is the 8-wide tree (RT_BUILD_WIDE): use rt_bvh_wide_dump");
    const DevBvh &b = which == 0 ? s->dev.scene : s->dev.lights;
    if (n_inner)
        *n_inner = s->dev_n_inner[which];
    if (n_tris)
        *n_tris = b.n_tris;
    if (root)
        *root = b.root;
    HIP_TRY(hipSetDevice(s->device));
    if (nodes64 && s->dev_n_inner[which])
        HIP_TRY(hipMemcpy(nodes64, b.nodes, sizeof(DevNode) * (size_t)s->dev_n_inner[which], hipMemcpyDeviceToHost));
    if (tris48 && b.n_tris)
        HIP_TRY(hipMemcpy(tris48, b.tris, sizeof(DevTri) * (size_t)b.n_tris, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_bvh_wide_dump(rt_scene *s, uint32_t *n_nodes, uint32_t *n_tris, uint32_t *depth, uint32_t *nodes80, uint32_t *tris48) {
    if (!s)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_bvh_wide_dump: null argument");
    if (s->group)
        return rt_bvh_wide_dump(rt::group_primary(s->group), n_nodes, n_tris, depth, nodes80, tris48);
    if (!s->wide_built)
        return rt::fail(RT_ERR_UNSUPPORTED, "rt_bvh_wide_dump: the scene was not built with RT_BUILD_WIDE");
    const DevBvh &b = s->dev.scene;
    if (n_nodes)
        *n_nodes = b.n_wide;
    if (n_tris)
        *n_tris = b.n_tris;
    if (depth)
        *depth = s->wide_depth;
    HIP_TRY(hipSetDevice(s->device));
    if (nodes80 && b.n_wide)
        HIP_TRY(hipMemcpy(nodes80, b.wide, sizeof(WideNode) * (size_t)b.n_wide, hipMemcpyDeviceToHost));
    if (tris48 && b.n_tris)
        HIP_TRY(hipMemcpy(tris48, b.tris, sizeof(DevTri) * (size_t)b.n_tris, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_build_times(const rt_scene *s, double *build_ms, double *upload_ms) {
    if (!s)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_build_times: null argument");
    if (s->group)
        return rt_build_times(rt::group_primary(s->group), build_ms, upload_ms);
    if (build_ms)
        *build_ms = s->build_ms;
    if (upload_ms)
        *upload_ms = s->build_upload_ms;
    return RT_OK;
}

extern "C" int rt_build_times_ex(const rt_scene *s, double *build_ms, double *upload_ms, double *wide_ms) {
    if (!s)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_build_times_ex: null argument");
    if (s->group)
        return rt_build_times_ex(rt::group_primary(s->group), build_ms, upload_ms, wide_ms);
    if (wide_ms)
        *wide_ms = s->wide_ms;
    return rt_build_times(s, build_ms, upload_ms);
}

// A device-built scene BVH has no host copy: rebuild the reference-style description (pre-order nodes with their OWN box,
// bvh.h:157-163) from what is in HBM, once, when a caller asks for it.
static int reconstruct_host_bvh(rt_scene *s) {
    const uint32_t n_inner = s->dev_n_inner[0], n_tris = s->dev.scene.n_tris;
    std::vector<DevNode> nodes(n_inner);
    std::vector<DevTri> tris(n_tris);
    HIP_TRY(hipSetDevice(s->device));
    if (n_inner)
        HIP_TRY(hipMemcpy(nodes.data(), s->dev.scene.nodes, sizeof(DevNode) * (size_t)n_inner, hipMemcpyDeviceToHost));
    if (n_tris)
        HIP_TRY(hipMemcpy(tris.data(), s->dev.scene.tris, sizeof(DevTri) * (size_t)n_tris, hipMemcpyDeviceToHost));
    rt::HostBvh &hb = s->rebuilt_bvh;
    hb.nodes.clear();
    hb.order.resize(n_tris);
    for (uint32_t k = 0; k < n_tris; ++k)
        hb.order[k] = tris[k].prim;
    hb.root = RT_NONE;
    if (s->dev.scene.root == RT_NONE)
        return RT_OK;
    struct Item {
        uint32_t ref, parent, side; // side: 0 root, 1 left, 2 right
        float lo[3], hi[3];
    };
    auto leaf_range = [&](uint32_t ref, uint32_t &b, uint32_t &e) {
        b = ref & RT_LEAF_BEGIN_MASK;
        const uint32_t cnt = RT_LEAF_CNT(ref);
        e = b + cnt;
        if (cnt == 0) // big leaf: walk the per-triangle flags
            for (e = b; e < n_tris && !(tris[e].flags & 1u); ++e) {
            }
        if (cnt == 0 && e < n_tris)
            ++e;
    };
    std::vector<Item> stack;
    Item r{};
    r.ref = s->dev.scene.root;
    r.parent = RT_NONE;
    for (int c = 0; c < 3; ++c) { // the root's own box is stored nowhere: union of its children's (or of its triangles)
        r.lo[c] = INFINITY;
        r.hi[c] = -INFINITY;
    }
    if (r.ref & RT_LEAF_FLAG) {
        uint32_t b, e;
        leaf_range(r.ref, b, e);
        for (uint32_t k = b; k < e; ++k)
            for (int v = 0; v < 3; ++v)
                for (int c = 0; c < 3; ++c) {
                    const float x = v == 0 ? tris[k].a[c] : (v == 1 ? tris[k].a[c] + tris[k].v[c] : tris[k].a[c] + tris[k].u[c]);
                    r.lo[c] = std::min(r.lo[c], x);
                    r.hi[c] = std::max(r.hi[c], x);
                }
    } else {
        const DevNode &nd = nodes[r.ref];
        for (int c = 0; c < 3; ++c) {
            r.lo[c] = std::min(nd.lmin[c], nd.rmin[c]);
            r.hi[c] = std::max(nd.lmax[c], nd.rmax[c]);
        }
    }
    stack.push_back(r);
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        const uint32_t idx = (uint32_t)hb.nodes.size();
        rt::HostNode hn{};
        std::memcpy(hn.lo, it.lo, 12);
        std::memcpy(hn.hi, it.hi, 12);
        hn.left = hn.right = RT_NONE;
        if (it.parent != RT_NONE)
            (it.side == 1 ? hb.nodes[it.parent].left : hb.nodes[it.parent].right) = idx;
        if (it.ref & RT_LEAF_FLAG) {
            leaf_range(it.ref, hn.obj_begin, hn.obj_end);
            hb.nodes.push_back(hn);
            continue;
        }
        hb.nodes.push_back(hn);
        const DevNode &nd = nodes[it.ref];
        Item l{}, rr{};
        l.ref = nd.left, l.parent = idx, l.side = 1;
        rr.ref = nd.right, rr.parent = idx, rr.side = 2;
        std::memcpy(l.lo, nd.lmin, 12);
        std::memcpy(l.hi, nd.lmax, 12);
        std::memcpy(rr.lo, nd.rmin, 12);
        std::memcpy(rr.hi, nd.rmax, 12);
        stack.push_back(rr); // pre-order: left subtree first
        stack.push_back(l);
    }
    hb.root = 0;
    return RT_OK;
}

extern "C" int rt_bvh_info(rt_scene *s, int which, uint32_t *n_nodes, uint32_t *n_objects, uint32_t *root, uint32_t *nodes_out,
                           uint32_t *order_out) {
    if (!s || which < 0 || which > 1)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_bvh_info: bad argument");
    if (s->group)
        return rt_bvh_info(rt::group_primary(s->group), which, n_nodes, n_objects, root, nodes_out, order_out);
    if (which == 0 && s->device_built && s->wide_built)
        return rt::fail(RT_ERR_UNSUPPORTED, "rt_bvh_info: the binary tree of a device-built wide scene is not kept");
    if (which == 0 && s->device_built && s->rebuilt_bvh.nodes.empty() && s->dev.scene.n_tris)
        if (int rc = reconstruct_host_bvh(s); rc != RT_OK)
            return rc;
    const rt::HostBvh &b = (which == 0 && s->device_built) ? s->rebuilt_bvh : s->prep->host_bvh[which];
    if (n_nodes)
        *n_nodes = (uint32_t)b.nodes.size();
    if (n_objects)
        *n_objects = (uint32_t)b.order.size();
    if (root)
        *root = b.root;
    if (nodes_out) {
        for (size_t i = 0; i < b.nodes.size(); ++i) {
            const rt::HostNode &nd = b.nodes[i];
            std::memcpy(nodes_out + 10 * i, nd.lo, 12);
            std::memcpy(nodes_out + 10 * i + 3, nd.hi, 12);
            nodes_out[10 * i + 6] = nd.left;
            nodes_out[10 * i + 7] = nd.right;
            nodes_out[10 * i + 8] = nd.obj_begin;
            nodes_out[10 * i + 9] = nd.obj_end;
        }
    }
    if (order_out && !b.order.empty())
        std::memcpy(order_out, b.order.data(), b.order.size() * sizeof(uint32_t));
    return RT_OK;
}

// Development aid (not part of include/rt_abi.h): copies and clears the -DRT_DIAG census words of the wavefront kernels.
extern "C" int rt_debug_census(rt_scene *s, unsigned long long *out32) {
    if (!s || !out32 || !s->wf_counters)
        return RT_ERR_INVALID_ARG;
    if (hipMemcpy(out32, s->wf_counters + 64, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
        return RT_ERR_HIP;
    (void)hipMemsetAsync(s->wf_counters + 64, 0, 32 * sizeof(unsigned long long), s->stream);
    (void)hipStreamSynchronize(s->stream);
    return RT_OK;
}

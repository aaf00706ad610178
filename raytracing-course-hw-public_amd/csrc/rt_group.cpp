// rt_group.cpp — one process, all GPUs of the node: scene replicas, interleaved image blocks, RCCL gather (rt_group.h).
//
// What this replaces in the reference: run_raytracer's thread pool (src/raytracer.h:636-665). There, hardware threads pull
// 256-pixel spans from one atomic counter and write disjoint pixels of one Image; here GPUs own interleaved pixel blocks
// (block b -> GPU b % G, fixed assignment: per-(pixel, sample) seeding makes the image independent of who renders what)
// and the only exchange is the gather of every GPU's finished blocks on GPU 0:
//     pack own blocks into a contiguous slab (one strided 2-D copy)  ->  ncclSend to rank 0 / grouped ncclRecv on rank 0
//     -> strided 2-D copy into the final image  ->  caller.
// 3 bytes per pixel with the device film (rt_render_rgb8), 12 with the float framebuffer: 3 - 50 MB in all, i.e. far
// below one xGMI link-second, so plain point-to-point sends to the root are the right collective.
//
// RCCL is loaded at run time (dlopen "librccl.so.1"): single-GPU users of librt_amd.so need no RCCL, and a process that
// already holds an RCCL (torch.distributed in bench.py) shares that one instead of loading a second copy.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_device_types.h"
#include "rt_error.h"
#include "rt_group.h"

namespace rt {

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle)
                break;
        }
        if (!r.handle) {
            r.error = std::string("RCCL not found: ") + dlerror();
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p && r.error.empty())
                r.error = std::string("RCCL symbol missing: ") + n;
            return p;
        };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

struct Replica {
    int device = 0;
    rt_scene *scene = nullptr;
    hipStream_t stream = nullptr; // pack / send / recv / unpack of this rank
    ncclComm_t comm = nullptr;
    char *image = nullptr; // full-size image of this rank (only its own blocks are written)
    char *slab = nullptr;  // its blocks, packed
    size_t image_cap = 0, slab_cap = 0;
};

// the blocks of rank r: b = r, r + G, ... < n_blocks; all full except possibly the image's last block
struct Blocks {
    uint64_t full = 0, tail = 0, first_tail_pixel = 0; // full blocks; pixels of a trailing partial block (0 = none)
    uint64_t pixels(uint64_t block) const { return full * block + tail; }
};
Blocks blocks_of(uint64_t n_pix, uint64_t block, uint32_t r, uint32_t G) {
    Blocks b;
    const uint64_t n_blocks = (n_pix + block - 1) / block;
    for (uint64_t k = r; k < n_blocks; k += G) {
        const uint64_t first = k * block, last = std::min(first + block, n_pix);
        if (last - first == block) {
            ++b.full;
        } else {
            b.tail = last - first;
            b.first_tail_pixel = first;
        }
    }
    return b;
}

} // namespace

struct Group {
    std::vector<Replica> ranks;
    bool use_rccl = true;       // false: RT_BUILD_GROUP_COPY (peer copies; rehearsal of the flow where RCCL cannot run)
    bool self_exchange = false; // RT_BUILD_GROUP_SELF_EXCHANGE: rank 0's own blocks also travel through ncclSend/ncclRecv (N = 1 test)
    char *recv = nullptr;       // on ranks[0].device: the other ranks' slabs
    size_t recv_cap = 0;
    bool comm_broken = false;   // an exchange failed and the communicators were aborted: later renders report RT_ERR_COMM at once
};

namespace {

int nccl_fail(const char *what, ncclResult_t r) {
    return fail(RT_ERR_COMM, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error"));
}
int hip_fail(const char *what, hipError_t e) { return fail(e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }

int ensure(char **buf, size_t *cap, size_t bytes) {
    if (*cap >= bytes)
        return RT_OK;
    if (*buf)
        (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess)
        return hip_fail("rt group: buffer", e);
    *buf = static_cast<char *>(p);
    *cap = bytes;
    return RT_OK;
}

// own blocks of rank r: image (strided) -> slab (packed), or back; es = bytes per pixel
hipError_t copy_blocks(char *slab, char *image, bool pack, const Blocks &b, uint64_t block, uint32_t r, uint32_t G, size_t es, hipStream_t st) {
    const size_t row = block * es;
    char *img0 = image + (size_t)r * row;
    if (b.full) {
        hipError_t e = pack ? hipMemcpy2DAsync(slab, row, img0, (size_t)G * row, row, b.full, hipMemcpyDeviceToDevice, st)
                            : hipMemcpy2DAsync(img0, (size_t)G * row, slab, row, row, b.full, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess)
            return e;
    }
    if (b.tail) {
        char *t_img = image + b.first_tail_pixel * es, *t_slab = slab + b.full * row;
        hipError_t e = pack ? hipMemcpyAsync(t_slab, t_img, b.tail * es, hipMemcpyDeviceToDevice, st) : hipMemcpyAsync(t_img, t_slab, b.tail * es, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

} // namespace

int group_size(const Group *g) { return g ? (int)g->ranks.size() : 0; }
rt_scene *group_primary(Group *g) { return g && !g->ranks.empty() ? g->ranks[0].scene : nullptr; }

void group_destroy(Group *g) {
    if (!g)
        return;
    for (Replica &r : g->ranks) {
        (void)hipSetDevice(r.device);
        if (r.comm && rccl().CommDestroy)
            (void)rccl().CommDestroy(r.comm);
        if (r.image)
            (void)hipFree(r.image);
        if (r.slab)
            (void)hipFree(r.slab);
        if (r.stream)
            (void)hipStreamDestroy(r.stream);
        if (r.scene)
            rt_destroy(r.scene);
    }
    if (g->recv && !g->ranks.empty()) {
        (void)hipSetDevice(g->ranks[0].device);
        (void)hipFree(g->recv);
    }
    delete g;
}

int group_create(const rt_scene_desc *desc, const int *devices, int n_devices, Group **out) {
    if (!desc || !devices || n_devices < 1 || !out)
        return fail(RT_ERR_INVALID_ARG, "rt group: bad argument");
    Group *g = new Group();
    g->use_rccl = !(desc->build_flags & RT_BUILD_GROUP_COPY);
    g->self_exchange = (desc->build_flags & RT_BUILD_GROUP_SELF_EXCHANGE) != 0;
    g->ranks.resize(n_devices);
    // The host half of rt_create (both reference-topology BVH builds, flattening / wide collapse, shading records, texture pool)
    // is done ONCE; every GPU then uploads the same arrays, one host thread per GPU (C5: 2.4 GB of 288 GB per replica).
    std::shared_ptr<const PreparedScene> prep;
    if (int rc = prepare(desc, &prep); rc != RT_OK) {
        delete g;
        return rc;
    }
    std::vector<int> rcs(n_devices, RT_OK);
    std::vector<std::string> errs(n_devices);
    std::vector<std::thread> th;
    for (int i = 0; i < n_devices; ++i) {
        g->ranks[i].device = devices[i];
        th.emplace_back([&, i] {
            rcs[i] = create_replica(desc, prep, devices[i], &g->ranks[i].scene);
            if (rcs[i] != RT_OK)
                errs[i] = rt_last_error();
            else if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&g->ranks[i].stream, hipStreamNonBlocking) != hipSuccess) {
                rcs[i] = RT_ERR_HIP;
                errs[i] = "rt group: stream creation failed";
            }
        });
    }
    for (auto &t : th)
        t.join();
    for (int i = 0; i < n_devices; ++i)
        if (rcs[i] != RT_OK) {
            const int rc = rcs[i];
            const std::string msg = "GPU " + std::to_string(devices[i]) + ": " + errs[i];
            group_destroy(g);
            return fail(rc, msg);
        }
    if (g->use_rccl) {
        Rccl &R = rccl();
        if (!R.error.empty()) {
            group_destroy(g);
            return fail(RT_ERR_COMM, R.error);
        }
        std::vector<ncclComm_t> comms(n_devices, nullptr);
        ncclResult_t nr = R.CommInitAll(comms.data(), n_devices, devices);
        if (nr != ncclSuccess) {
            group_destroy(g);
            return nccl_fail("ncclCommInitAll", nr);
        }
        for (int i = 0; i < n_devices; ++i)
            g->ranks[i].comm = comms[i];
    }
    *out = g;
    return RT_OK;
}

int group_render(Group *g, const rt_params *p, float *fb, uint8_t *rgb8, rt_stats *stats) {
    if (!g || !p || (!fb && !rgb8))
        return fail(RT_ERR_INVALID_ARG, "rt_render: null argument");
    if (p->shard_count > 1)
        return fail(RT_ERR_INVALID_ARG, "rt_render: a multi-GPU scene shards the image itself; shard_count must be 0 or 1");
    if (p->width == 0 || p->height == 0 || (uint64_t)p->width * p->height >= 0x7FFFFFFFull)
        return fail(RT_ERR_INVALID_ARG, "Illegal image size" + std::to_string(p->width) + "x" + std::to_string(p->height)); // image.h:26
    const auto wall0 = std::chrono::steady_clock::now();
    const uint32_t G = (uint32_t)g->ranks.size();
    const uint64_t n_pix = (uint64_t)p->width * p->height;
    const size_t es = rgb8 ? 3 : 12;
    // 8 image rows per block (cost per row is very non-uniform: interleave finely); whole 256-pixel spans in reference-RNG
    // mode, whose seed is the span index (config.h:13, raytracer.h:648)
    uint64_t block = p->shard_block ? p->shard_block : 8ull * p->width;
    if (p->rng_mode == RT_RNG_REFERENCE)
        block = (block + RT_SPAN - 1) / RT_SPAN * RT_SPAN;
    const bool device_out = (p->flags & RT_FLAG_DEVICE_FB) != 0;
    char *caller = rgb8 ? reinterpret_cast<char *>(rgb8) : reinterpret_cast<char *>(fb);
    const bool exchange0 = g->self_exchange && g->use_rccl; // rank 0's own blocks go through RCCL too

    // ---- buffers
    std::vector<Blocks> blk(G);
    std::vector<size_t> recv_off(G, 0);
    size_t recv_bytes = 0;
    for (uint32_t r = 0; r < G; ++r) {
        blk[r] = blocks_of(n_pix, block, r, G);
        if (r > 0 || exchange0) {
            recv_off[r] = recv_bytes;
            recv_bytes += blk[r].pixels(block) * es;
        }
    }
    for (uint32_t r = 0; r < G; ++r) {
        Replica &R = g->ranks[r];
        if (hipError_t e = hipSetDevice(R.device); e != hipSuccess)
            return hip_fail("hipSetDevice", e);
        const bool needs_image = r > 0 || !device_out || exchange0;
        if (int rc = needs_image ? ensure(&R.image, &R.image_cap, n_pix * es) : RT_OK; rc != RT_OK)
            return rc;
        if (r > 0 || exchange0)
            if (int rc = ensure(&R.slab, &R.slab_cap, blk[r].pixels(block) * es); rc != RT_OK)
                return rc;
    }
    (void)hipSetDevice(g->ranks[0].device);
    if (int rc = ensure(&g->recv, &g->recv_cap, recv_bytes); rc != RT_OK)
        return rc;
    // where rank 0 assembles the final image: the caller's device buffer, or its own
    char *final_img = device_out ? caller : g->ranks[0].image;
    // where rank 0 renders: straight into the final image, unless its blocks are to be exchanged as well
    char *render0 = exchange0 ? g->ranks[0].image : final_img;
    if (exchange0 && device_out && render0 == final_img)
        return fail(RT_ERR_INVALID_ARG, "rt group: internal buffer aliasing");

    // ---- phase 1: every GPU renders its blocks (one host thread per GPU, as one std::thread per core in raytracer.h:636-662)
    std::vector<int> rcs(G, RT_OK);
    std::vector<std::string> errs(G);
    std::vector<rt_stats> sts(G);
    {
        std::mutex progress_mutex; // rt_params.progress: one call per GPU that finished, serialised
        uint32_t finished = 0;
        std::vector<std::thread> th;
        for (uint32_t r = 0; r < G; ++r)
            th.emplace_back([&, r] {
                rt_params q = *p;
                q.shard_index = r;
                q.shard_count = G;
                q.shard_block = (uint32_t)block;
                q.flags |= RT_FLAG_DEVICE_FB;
                q.progress = nullptr;
                char *dst = r == 0 ? render0 : g->ranks[r].image;
                rcs[r] = rgb8 ? rt_render_rgb8(g->ranks[r].scene, &q, reinterpret_cast<uint8_t *>(dst), &sts[r])
                              : rt_render(g->ranks[r].scene, &q, reinterpret_cast<float *>(dst), &sts[r]);
                if (rcs[r] != RT_OK)
                    errs[r] = rt_last_error();
                else if (p->progress) {
                    std::lock_guard<std::mutex> lock(progress_mutex);
                    p->progress(++finished, G, p->progress_user);
                }
            });
        for (auto &t : th)
            t.join();
    }
    for (uint32_t r = 0; r < G; ++r)
        if (rcs[r] != RT_OK)
            return fail(rcs[r], "GPU " + std::to_string(g->ranks[r].device) + ": " + errs[r]);

    // ---- phase 2: gather on rank 0. Every rank packs and sends on its own stream; rank 0 posts all receives in one group.
    // Nobody may be left waiting: (1) whether the exchange happens at all is decided BEFORE the threads start (a rank that
    // cannot even select its device would never post its send); (2) a rank whose RCCL call fails aborts EVERY communicator of
    // the group (ncclCommAbort), which completes the peers' pending operations with an error instead of letting their
    // hipStreamSynchronize wait for a transfer that will never be matched; the group is unusable afterwards (RT_ERR_COMM).
    if (g->use_rccl && g->comm_broken)
        return fail(RT_ERR_COMM, "rt group: the communicator was aborted by an earlier failed exchange; create the scene again");
    for (uint32_t r = 0; r < G; ++r)
        if (hipError_t e = hipSetDevice(g->ranks[r].device); e != hipSuccess)
            return hip_fail("rt group: hipSetDevice before the exchange", e);
    // The rank threads work on a COPY of the communicator handles taken here, before any of them starts; a failing rank sets `broken`
    // and aborts every communicator of that copy once (abort completes the peers' pending operations with an error). Replica::comm is
    // only written after the join, so no thread ever reads a handle another one is clearing, and a rank that has not posted yet when the
    // flag goes up skips its RCCL calls instead of posting on an aborted communicator.
    std::vector<ncclComm_t> comms(G, nullptr);
    for (uint32_t r = 0; r < G; ++r)
        comms[r] = g->ranks[r].comm;
    std::atomic<bool> broken{false};
    std::mutex abort_mutex;
    auto abort_all = [&]() {
        std::lock_guard<std::mutex> lock(abort_mutex);
        if (broken.exchange(true) || !rccl().CommAbort)
            return;
        for (ncclComm_t c : comms)
            if (c)
                (void)rccl().CommAbort(c);
    };
    {
        std::vector<std::thread> th;
        for (uint32_t r = 0; r < G; ++r)
            th.emplace_back([&, r] {
                Replica &R = g->ranks[r];
                Rccl &N = rccl();
                auto hip_try = [&](hipError_t e, const char *what) {
                    if (e != hipSuccess && rcs[r] == RT_OK) {
                        rcs[r] = RT_ERR_HIP;
                        errs[r] = std::string(what) + ": " + hipGetErrorString(e);
                    }
                    return e == hipSuccess;
                };
                bool comm_failed = false;
                auto nccl_try = [&](ncclResult_t e, const char *what) {
                    if (e != ncclSuccess) {
                        comm_failed = true;
                        if (rcs[r] == RT_OK) {
                            rcs[r] = RT_ERR_COMM;
                            errs[r] = std::string(what) + ": " + N.GetErrorString(e);
                        }
                    }
                    return e == ncclSuccess;
                };
                ncclComm_t comm = comms[r];
                (void)hip_try(hipSetDevice(R.device), "hipSetDevice"); // verified for every rank just above; the exchange is posted regardless
                const bool sends = r > 0 || exchange0;
                const size_t bytes = blk[r].pixels(block) * es;
                if (sends && bytes)
                    hip_try(copy_blocks(R.slab, r == 0 ? render0 : R.image, true, blk[r], block, r, G, es, R.stream), "pack");
                if (g->use_rccl && broken.load()) {
                    if (rcs[r] == RT_OK) {
                        rcs[r] = RT_ERR_COMM;
                        errs[r] = "the exchange was aborted by another rank's failure";
                    }
                } else if (g->use_rccl) {
                    // (a rank that failed above still takes part in the exchange: nobody may be left waiting in ncclRecv)
                    if (r == 0) {
                        nccl_try(N.GroupStart(), "ncclGroupStart");
                        if (exchange0 && bytes)
                            nccl_try(N.Send(R.slab, bytes, ncclUint8, 0, comm, R.stream), "ncclSend");
                        for (uint32_t q = exchange0 ? 0 : 1; q < G; ++q)
                            if (const size_t qb = blk[q].pixels(block) * es)
                                nccl_try(N.Recv(g->recv + recv_off[q], qb, ncclUint8, (int)q, comm, R.stream), "ncclRecv");
                        nccl_try(N.GroupEnd(), "ncclGroupEnd");
                    } else if (bytes) {
                        nccl_try(N.Send(R.slab, bytes, ncclUint8, 0, comm, R.stream), "ncclSend");
                    }
                    if (comm_failed)
                        abort_all(); // the peers' matching operations would never complete
                }
                if (!hip_try(hipStreamSynchronize(R.stream), "gather stream") && g->use_rccl)
                    abort_all();
            });
        for (auto &t : th)
            t.join();
    }
    if (broken.load()) { // the handles are gone (ncclCommAbort frees them): later renders report RT_ERR_COMM at once
        g->comm_broken = true;
        for (Replica &R : g->ranks)
            R.comm = nullptr;
    }
    for (uint32_t r = 0; r < G; ++r)
        if (rcs[r] != RT_OK)
            return fail(rcs[r], "GPU " + std::to_string(g->ranks[r].device) + " (gather): " + errs[r]);
    Replica &R0 = g->ranks[0];
    if (hipError_t e = hipSetDevice(R0.device); e != hipSuccess)
        return hip_fail("hipSetDevice", e);
    if (!g->use_rccl) // rehearsal transport: peer copies of the packed slabs (same packing, same unpacking), ordered on rank
                      // 0's stream with the unpacking below (a device-to-device hipMemcpyPeer may return before it is done)
        for (uint32_t q = 1; q < G; ++q)
            if (const size_t qb = blk[q].pixels(block) * es)
                if (hipError_t e = hipMemcpyPeerAsync(g->recv + recv_off[q], R0.device, g->ranks[q].slab, g->ranks[q].device, qb, R0.stream); e != hipSuccess)
                    return hip_fail("hipMemcpyPeerAsync", e);
    for (uint32_t q = exchange0 ? 0 : 1; q < G; ++q)
        if (blk[q].pixels(block))
            if (hipError_t e = copy_blocks(g->recv + recv_off[q], final_img, false, blk[q], block, q, G, es, R0.stream); e != hipSuccess)
                return hip_fail("unpack", e);
    if (!device_out)
        if (hipError_t e = hipMemcpyAsync(caller, final_img, n_pix * es, hipMemcpyDeviceToHost, R0.stream); e != hipSuccess)
            return hip_fail("image copy", e);
    if (hipError_t e = hipStreamSynchronize(R0.stream); e != hipSuccess)
        return hip_fail("gather", e);

    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (uint32_t r = 0; r < G; ++r) {
            const rt_stats &s = sts[r];
            stats->samples += s.samples;
            stats->casts += s.casts;
            stats->nodes_visited += s.nodes_visited;
            stats->box_tests += s.box_tests;
            stats->tri_tests += s.tri_tests;
            stats->shaded_hits += s.shaded_hits;
            stats->light_queries += s.light_queries;
            stats->light_nodes += s.light_nodes;
            stats->light_box_tests += s.light_box_tests;
            stats->light_tri_tests += s.light_tri_tests;
            stats->light_hits += s.light_hits;
            stats->texel_fetches += s.texel_fetches;
            stats->kernel_ms = std::max(stats->kernel_ms, s.kernel_ms); // GPUs run concurrently: the slowest one counts
            stats->dominant_ms = std::max(stats->dominant_ms, s.dominant_ms);
            stats->dominant_launches = std::max(stats->dominant_launches, s.dominant_launches);
            stats->packet_lanes_x100 = std::max(stats->packet_lanes_x100, s.packet_lanes_x100);
            stats->passes = std::max(stats->passes, s.passes);
            stats->packet_passes = std::max(stats->packet_passes, s.packet_passes);
        }
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    }
    return RT_OK;
}

} // namespace rt

// rt_bvh_device.hip — BVH construction ON THE GPU: the production builder for big scenes.
//
// The reference builds its BVH on one host thread with a full std::sort per node and a surface-area sweep
// (src/bvh.h:268-313, 323-366): O(n log^2 n), about half a second per 2*10^5 triangles and minutes at 10^7. The host
// builder of bvh_build.cpp reproduces that topology bit for bit (needed for the parity contract: event counters and exact
// ties follow the tree) and stays the default. This file is the other mode (rt_scene_desc.build_flags &
// RT_BUILD_DEVICE_LBVH): a linear BVH built entirely on the device in a few tens of milliseconds for 10^7 triangles,
// written straight into the SAME HBM layout the traversal kernels read (DevNode: both children's boxes + child refs,
// DevTri in leaf order, DevAttr in the same order), so nothing downstream changes.
//
//   1. scene bounds              block reduction + ordered-integer atomics
//   2. 30-bit Morton code of every triangle's centre (triangle::center, geometry.h:485-487), radix sort (rocPRIM)
//   3. leaves = runs of LEAF consecutive sorted triangles (default 1; the reference's leaves hold 3.9 on average,
//      bvh.h:343-346); per leaf: DevTri / DevAttr records in sorted order, exact AABB of its vertices, 64-bit key = Morton
//      of its first triangle : leaf index (unique, sorted)
//   4. Karras 2012 radix tree over the leaf keys: every inner node finds its own range and split independently
//   5. bottom-up refit: each leaf walks to the root; the second thread to arrive at a node has both child boxes, writes
//      them into that node's DevNode and carries the union upwards (one agent-scope fence + atomic per hand-off)
//
// Boxes are exact (min / max of vertex coordinates, no arithmetic), so the traversal's exactness argument
// (div_exact_fast preconditions, rt_device_lib.h) holds unchanged. What changes is the TOPOLOGY: rays still find the
// closest hit (same t, bit for bit; tests/test_gpu_bvh_device.py), but equal-t ties may resolve to another triangle
// and the event counters differ from the reference's, so this mode is not the parity mode.
#include <hip/hip_runtime.h>

#include <chrono>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "rt_bvh_device.h"
#include "rt_kernels.h"

namespace {

// Triangles per leaf. Measured (tools/lbvh_probe.py, profiles/r02_lbvh.txt): 1 renders fastest at both scene sizes
// (262 k triangles: 83 ms against 107 ms on the reference's tree; 10^7: 166 against 141 ms), 2 and 4 are slower (loose leaf
// boxes along the Morton curve), so the default is one triangle per leaf; RT_LBVH_LEAF (1..8) overrides it for experiments.
constexpr uint32_t LEAF_TRIS_DEFAULT = 1;

__device__ __forceinline__ uint32_t enc_f(float f) { // order-preserving float -> uint
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float dec_f(uint32_t e) {
    const uint32_t b = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(b);
#else
    std::memcpy(&f, &b, 4);
#endif
    return f;
}

// ---- 1. bounds of all vertices: bounds[0..2] = min (encoded), bounds[3..5] = max
__global__ __launch_bounds__(256) void k_bounds(const float *__restrict__ pos, uint32_t n, uint32_t *bounds) {
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *p = pos + 9ull * i;
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                lo[c] = fminf(lo[c], p[3 * v + c]);
                hi[c] = fmaxf(hi[c], p[3 * v + c]);
            }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[c] = fminf(lo[c], __shfl_down(lo[c], off));
            hi[c] = fmaxf(hi[c], __shfl_down(hi[c], off));
        }
    }
    if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            atomicMin(bounds + c, enc_f(lo[c]));
            atomicMax(bounds + 3 + c, enc_f(hi[c]));
        }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) { // 10 bits -> every third bit
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// ---- 2. Morton keys of triangle centres
__global__ __launch_bounds__(256) void k_keys(const float *__restrict__ pos, uint32_t n, const uint32_t *bounds, uint32_t *keys, uint32_t *vals) {
    const float lo[3] = {dec_f(bounds[0]), dec_f(bounds[1]), dec_f(bounds[2])};
    float inv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ext = dec_f(bounds[3 + c]) - lo[c];
        inv[c] = ext > 0.0f ? 1024.0f / ext : 0.0f;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *p = pos + 9ull * i;
        uint32_t q[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ctr = (p[c] + p[3 + c] + p[6 + c]) / 3.0f;
            q[c] = (uint32_t)fminf(fmaxf((ctr - lo[c]) * inv[c], 0.0f), 1023.0f); // NaN -> 0
        }
        keys[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
        vals[i] = i;
    }
}

struct BuildArrays {
    const float *pos, *nrm, *tan, *uv;
    const uint32_t *mat;
    const uint32_t *keys_sorted, *prims_sorted;
    uint32_t n, n_leaves, leaf_tris;
    DevTri *tris;
    DevAttr *attrs;
    DevNode *nodes;
    unsigned long long *leaf_keys;
    float *leaf_box; // [n_leaves][6]
    float *node_box; // [n_leaves - 1][6]
    uint32_t *leaf_parent, *node_parent;
    uint32_t *arrived; // [n_leaves - 1]
    uint32_t *fast_bad; // set to 1 if a coordinate violates the div_exact_fast range (rt_device_lib.h)
};

__device__ __forceinline__ bool coord_fast_ok(float c) {
    const float m = __builtin_fabsf(c);
    return (c == 0.0f) | ((m >= 7.275957614183426e-12f) & (m <= 1099511627776.0f));
}

// ---- 3. leaves: records in sorted order, exact boxes, keys
__global__ __launch_bounds__(256) void k_leaves(const BuildArrays A) {
    for (uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x; leaf < A.n_leaves; leaf += gridDim.x * blockDim.x) {
        const uint32_t k0 = leaf * A.leaf_tris, k1 = min(k0 + A.leaf_tris, A.n);
        float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        bool ok = true;
        for (uint32_t k = k0; k < k1; ++k) {
            const uint32_t prim = A.prims_sorted[k];
            const float *p = A.pos + 9ull * prim;
            DevTri t;
            DevAttr at;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                t.a[c] = p[c];
                t.v[c] = p[3 + c] - p[c]; // triangle::v geometry.h:473
                t.u[c] = p[6 + c] - p[c]; // triangle::u geometry.h:475
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    lo[c] = fminf(lo[c], p[3 * v + c]);
                    hi[c] = fmaxf(hi[c], p[3 * v + c]);
                    ok &= coord_fast_ok(p[3 * v + c]);
                }
            }
            t.prim = prim;
            t.flags = (k == k1 - 1 ? 1u : 0u) | (k == k0 ? 2u : 0u);
            t.pad = 0;
            A.tris[k] = t;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                at.n[j] = A.nrm[9ull * prim + j];
                at.tg[j] = A.tan[9ull * prim + j];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
                at.uv[j] = A.uv[6ull * prim + j];
            // base_normal() = norm(crs(v, u)) (geometry.h:477-479, 648-650), same float operations as the host path
            const float cx = t.v[1] * t.u[2] - t.v[2] * t.u[1], cy = t.v[2] * t.u[0] - t.v[0] * t.u[2], cz = t.v[0] * t.u[1] - t.v[1] * t.u[0];
            const float l = __builtin_sqrtf(cx * cx + cy * cy + cz * cz);
            at.gn[0] = cx / l;
            at.gn[1] = cy / l;
            at.gn[2] = cz / l;
            at.material = A.mat[prim];
            at.pad[0] = at.pad[1] = at.pad[2] = at.pad[3] = 0;
            A.attrs[k] = at;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            A.leaf_box[6ull * leaf + c] = lo[c];
            A.leaf_box[6ull * leaf + 3 + c] = hi[c];
        }
        A.leaf_keys[leaf] = ((unsigned long long)A.keys_sorted[k0] << 32) | (unsigned long long)leaf;
        if (!ok)
            *A.fast_bad = 1u;
    }
}

__device__ __forceinline__ uint32_t leaf_ref(const BuildArrays &A, uint32_t leaf) {
    const uint32_t k0 = leaf * A.leaf_tris, cnt = min(A.leaf_tris, A.n - k0);
    return RT_LEAF_FLAG | (cnt << 27) | k0;
}

// ---- 4. Karras 2012: inner node i of the radix tree over the sorted, unique leaf keys
__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n)
        return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}
__global__ __launch_bounds__(256) void k_radix_tree(const BuildArrays A) {
    const int n = (int)A.n_leaves;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
        const unsigned long long *keys = A.leaf_keys;
        const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
        const int dmin = delta(keys, n, i, i - d);
        int lmax = 2;
        while (delta(keys, n, i, i + lmax * d) > dmin)
            lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (delta(keys, n, i, i + (l + t) * d) > dmin)
                l += t;
        const int j = i + l * d;
        const int dnode = delta(keys, n, i, j);
        int s = 0;
        for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
            if (delta(keys, n, i, i + (s + t) * d) > dnode)
                s += t;
            if (t <= 1)
                break;
        }
        const int gamma = i + s * d + (d < 0 ? -1 : 0);
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        DevNode &nd = A.nodes[i];
        if (lo == gamma) {
            nd.left = leaf_ref(A, (uint32_t)gamma);
            A.leaf_parent[gamma] = (uint32_t)i;
        } else {
            nd.left = (uint32_t)gamma;
            A.node_parent[gamma] = (uint32_t)i;
        }
        if (hi == gamma + 1) {
            nd.right = leaf_ref(A, (uint32_t)(gamma + 1));
            A.leaf_parent[gamma + 1] = (uint32_t)i;
        } else {
            nd.right = (uint32_t)(gamma + 1);
            A.node_parent[gamma + 1] = (uint32_t)i;
        }
        nd.pad[0] = nd.pad[1] = 0;
        if (i == 0)
            A.node_parent[0] = RT_NONE;
    }
}

// ---- 5. refit: the second arrival at a node owns it. Hand-off between workgroups: the first arrival has stored its
// subtree's box, fenced (agent-scope release) and bumped the counter; the second one sees counter == 1, fences
// (agent-scope acquire) and reads that box (MI355X_MICROARCH.md, inter-workgroup visibility).
__global__ __launch_bounds__(256) void k_refit(const BuildArrays A) {
    for (uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x; leaf < A.n_leaves; leaf += gridDim.x * blockDim.x) {
        float box[6];
#pragma unroll
        for (int c = 0; c < 6; ++c)
            box[c] = A.leaf_box[6ull * leaf + c];
        uint32_t child_is_leaf = 1u, child = leaf;
        uint32_t node = A.leaf_parent[leaf];
        while (node != RT_NONE) {
            // publish this subtree's box where the sibling's thread will look for it
            float *mine = child_is_leaf ? A.leaf_box + 6ull * child : A.node_box + 6ull * child;
            if (!child_is_leaf) {
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    mine[c] = box[c];
            }
            __threadfence();
            const uint32_t before = atomicAdd(A.arrived + node, 1u);
            if (before == 0u)
                break; // first arrival: the sibling's thread will finish this node
            __threadfence();
            DevNode &nd = A.nodes[node];
            const uint32_t lref = nd.left, rref = nd.right;
            const float *lb = (lref & RT_LEAF_FLAG) ? A.leaf_box + 6ull * ((lref & RT_LEAF_BEGIN_MASK) / A.leaf_tris) : A.node_box + 6ull * lref;
            const float *rb = (rref & RT_LEAF_FLAG) ? A.leaf_box + 6ull * ((rref & RT_LEAF_BEGIN_MASK) / A.leaf_tris) : A.node_box + 6ull * rref;
            float l6[6], r6[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                l6[c] = __hip_atomic_load(lb + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                r6[c] = __hip_atomic_load(rb + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                nd.lmin[c] = l6[c];
                nd.lmax[c] = l6[3 + c];
                nd.rmin[c] = r6[c];
                nd.rmax[c] = r6[3 + c];
                box[c] = fminf(l6[c], r6[c]);
                box[3 + c] = fmaxf(l6[3 + c], r6[3 + c]);
            }
            child_is_leaf = 0u;
            child = node;
            node = A.node_parent[node];
        }
    }
}

struct Tmp { // device allocations of the build, freed on every return path
    std::vector<void *> ptrs;
    template <class T> hipError_t alloc(T **p, size_t count) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess)
            ptrs.push_back(q);
        *p = static_cast<T *>(q);
        return e;
    }
    ~Tmp() {
        for (void *p : ptrs)
            (void)hipFree(p);
    }
};

} // namespace

namespace rt {

#define BUILD_TRY(expr)            \
    do {                           \
        hipError_t e_ = (expr);    \
        if (e_ != hipSuccess) {    \
            if (err)               \
                *err = #expr;      \
            return e_;             \
        }                          \
    } while (0)

hipError_t build_bvh_device(const rt_scene_desc *d, hipStream_t stream, DeviceBvh *out, const char **err) {
    const uint32_t n = d->n_triangles;
    *out = DeviceBvh{};
    out->root = RT_NONE;
    out->fast_ok = true;
    if (n == 0)
        return hipSuccess;
    const auto t0 = std::chrono::steady_clock::now();
    Tmp tmp;
    float *pos, *nrm, *tan, *uv;
    uint32_t *mat, *keys[2], *vals[2], *bounds, *leaf_parent, *node_parent, *arrived, *fast_bad;
    uint32_t leaf_tris = LEAF_TRIS_DEFAULT;
    if (const char *e = std::getenv("RT_LBVH_LEAF"))
        leaf_tris = (uint32_t)std::min(8, std::max(1, std::atoi(e)));
    const uint32_t n_leaves = (n + leaf_tris - 1) / leaf_tris;
    BUILD_TRY(tmp.alloc(&pos, 9ull * n));
    BUILD_TRY(tmp.alloc(&nrm, 9ull * n));
    BUILD_TRY(tmp.alloc(&tan, 9ull * n));
    BUILD_TRY(tmp.alloc(&uv, 6ull * n));
    BUILD_TRY(tmp.alloc(&mat, (size_t)n));
    BUILD_TRY(hipMemcpyAsync(pos, d->positions, 36ull * n, hipMemcpyHostToDevice, stream));
    BUILD_TRY(hipMemcpyAsync(nrm, d->normals, 36ull * n, hipMemcpyHostToDevice, stream));
    BUILD_TRY(hipMemcpyAsync(tan, d->tangents, 36ull * n, hipMemcpyHostToDevice, stream));
    BUILD_TRY(hipMemcpyAsync(uv, d->texcoords, 24ull * n, hipMemcpyHostToDevice, stream));
    BUILD_TRY(hipMemcpyAsync(mat, d->material_ids, 4ull * n, hipMemcpyHostToDevice, stream));
    for (int k = 0; k < 2; ++k) {
        BUILD_TRY(tmp.alloc(&keys[k], (size_t)n));
        BUILD_TRY(tmp.alloc(&vals[k], (size_t)n));
    }
    BUILD_TRY(tmp.alloc(&bounds, (size_t)8));
    const uint32_t init_bounds[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
    BUILD_TRY(hipMemcpyAsync(bounds, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, stream));
    fast_bad = bounds + 6;
    BUILD_TRY(hipStreamSynchronize(stream)); // `init_bounds` is a local; uploads done = start of the device build proper
    const auto t1 = std::chrono::steady_clock::now();

    const int blocks = (int)std::min<uint64_t>(((uint64_t)n + 255) / 256, 256u * 16u);
    BUILD_TRY(RT_LAUNCH_CHECKED(k_bounds, dim3(blocks), dim3(256), 0, stream, pos, n, bounds));
    BUILD_TRY(RT_LAUNCH_CHECKED(k_keys, dim3(blocks), dim3(256), 0, stream, pos, n, bounds, keys[0], vals[0]));
    size_t sort_bytes = 0;
    BUILD_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys[0], keys[1], vals[0], vals[1], (size_t)n, 0u, 30u, stream));
    char *sort_tmp;
    BUILD_TRY(tmp.alloc(&sort_tmp, sort_bytes));
    BUILD_TRY(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys[0], keys[1], vals[0], vals[1], (size_t)n, 0u, 30u, stream));

    // outputs (owned by the caller on success)
    std::vector<void *> outs;
    auto out_alloc = [&](void **p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes ? bytes : 16);
        if (e == hipSuccess)
            outs.push_back(*p);
        return e;
    };
    struct OutGuard {
        std::vector<void *> &v;
        bool keep = false;
        ~OutGuard() {
            if (!keep)
                for (void *p : v)
                    (void)hipFree(p);
        }
    } guard{outs};
    BuildArrays A{};
    A.pos = pos, A.nrm = nrm, A.tan = tan, A.uv = uv, A.mat = mat;
    A.keys_sorted = keys[1], A.prims_sorted = vals[1];
    A.n = n, A.n_leaves = n_leaves, A.leaf_tris = leaf_tris;
    BUILD_TRY(out_alloc((void **)&A.tris, sizeof(DevTri) * (size_t)n));
    BUILD_TRY(out_alloc((void **)&A.attrs, sizeof(DevAttr) * (size_t)n));
    BUILD_TRY(out_alloc((void **)&A.nodes, sizeof(DevNode) * (size_t)(n_leaves > 1 ? n_leaves - 1 : 1)));
    BUILD_TRY(tmp.alloc(&A.leaf_keys, (size_t)n_leaves));
    BUILD_TRY(tmp.alloc(&A.leaf_box, 6ull * n_leaves));
    BUILD_TRY(tmp.alloc(&A.node_box, 6ull * n_leaves));
    BUILD_TRY(tmp.alloc(&leaf_parent, (size_t)n_leaves));
    BUILD_TRY(tmp.alloc(&node_parent, (size_t)n_leaves));
    BUILD_TRY(tmp.alloc(&arrived, (size_t)n_leaves));
    A.leaf_parent = leaf_parent, A.node_parent = node_parent, A.arrived = arrived, A.fast_bad = fast_bad;
    BUILD_TRY(hipMemsetAsync(arrived, 0, 4ull * n_leaves, stream));
    BUILD_TRY(hipMemsetAsync(leaf_parent, 0xFF, 4ull * n_leaves, stream)); // RT_NONE: a single leaf has no parent
    const int lblocks = (int)std::min<uint64_t>(((uint64_t)n_leaves + 255) / 256, 256u * 16u);
    BUILD_TRY(RT_LAUNCH_CHECKED(k_leaves, dim3(lblocks), dim3(256), 0, stream, A));
    if (n_leaves > 1) {
        BUILD_TRY(RT_LAUNCH_CHECKED(k_radix_tree, dim3(lblocks), dim3(256), 0, stream, A));
        BUILD_TRY(RT_LAUNCH_CHECKED(k_refit, dim3(lblocks), dim3(256), 0, stream, A));
    }
    uint32_t h_bounds[8];
    BUILD_TRY(hipMemcpyAsync(h_bounds, bounds, sizeof(h_bounds), hipMemcpyDeviceToHost, stream));
    BUILD_TRY(hipStreamSynchronize(stream));
    const auto t2 = std::chrono::steady_clock::now();

    guard.keep = true;
    out->nodes = A.nodes;
    out->tris = A.tris;
    out->attrs = A.attrs;
    out->n_inner = n_leaves > 1 ? n_leaves - 1 : 0;
    out->n_tris = n;
    out->root = n_leaves > 1 ? 0u : (RT_LEAF_FLAG | (n << 27) | 0u);
    out->fast_ok = h_bounds[6] == 0u;
    for (int c = 0; c < 3; ++c) {
        out->lo[c] = dec_f(h_bounds[c]);
        out->hi[c] = dec_f(h_bounds[3 + c]);
    }
    out->upload_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    out->build_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    return hipSuccess;
}

} // namespace rt

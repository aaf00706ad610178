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
#include "wide_grid.h"
#include "rt_kernels.h"

namespace {

// Triangles per leaf. Measured (tools/lbvh_probe.py, profiles/r02_lbvh.txt): 1 renders fastest at both scene sizes
// (262 k triangles: 83 ms against 107 ms on the reference's tree; 10^7: 166 against 141 ms), 2 and 4 are slower (loose leaf
// boxes along the Morton curve), so the default is one triangle per leaf; rt_build_options.lbvh_leaf_tris (1..8) overrides it for experiments.
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
    // wide collapse (RT_BUILD_WIDE on top of the device tree): the dynamic program of wide_build.cpp, filled by k_refit
    float *dp_cost;                // [n_leaves - 1][7]: C(n, i), i = 1..7; null = no collapse wanted
    unsigned long long *dp_dec;    // [n_leaves - 1]: ksplit(j = 2..8) 3 bits each from bit 0, eff(i = 1..7) 3 bits each from bit 21, as_leaf bit 42
    uint32_t *dp_ntris;            // [n_leaves - 1]: triangles below
    float cost_node, cost_tri;
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

__device__ __forceinline__ float box_area6(const float *b) { // surface area of {lo.xyz, hi.xyz}; 0 for an inverted / NaN box
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return (dx >= 0.0f && dy >= 0.0f && dz >= 0.0f) ? 2.0f * (dx * dy + dy * dz + dz * dx) : 0.0f;
}

// The wide collapse's dynamic program for one binary node (wide_build.cpp step 2; children are complete by construction):
// C(n, i) = cheapest representation of the subtree as at most i roots, i = 1..7, and the decisions that reach it.
__device__ __forceinline__ void dp_node(const BuildArrays &A, uint32_t node, uint32_t lref, uint32_t rref, const float *l6, const float *r6, const float *box) {
    const float INF = __builtin_inff();
    float cl[7], cr[7];
    uint32_t nl = 1u, nr = 1u;
    if (lref & RT_LEAF_FLAG) {
        const float a = box_area6(l6) * A.cost_tri;
#pragma unroll
        for (int i = 0; i < 7; ++i)
            cl[i] = a;
    } else {
#pragma unroll
        for (int i = 0; i < 7; ++i)
            cl[i] = __hip_atomic_load(A.dp_cost + 7ull * lref + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nl = __hip_atomic_load(A.dp_ntris + lref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (rref & RT_LEAF_FLAG) {
        const float a = box_area6(r6) * A.cost_tri;
#pragma unroll
        for (int i = 0; i < 7; ++i)
            cr[i] = a;
    } else {
#pragma unroll
        for (int i = 0; i < 7; ++i)
            cr[i] = __hip_atomic_load(A.dp_cost + 7ull * rref + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nr = __hip_atomic_load(A.dp_ntris + rref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float area = box_area6(box);
    const uint32_t nt = nl + nr;
    unsigned long long dec = 0ull;
    float dist[9];
#pragma unroll
    for (int j = 2; j <= 8; ++j) {
        float best = INF;
        int bk = 1;
#pragma unroll
        for (int k = 1; k < j; ++k) {
            if (k > 7 || j - k > 7)
                continue;
            const float c = cl[k - 1] + cr[j - k - 1];
            if (c < best) {
                best = c;
                bk = k;
            }
        }
        dist[j] = best;
        dec |= (unsigned long long)bk << (3 * (j - 2));
    }
    const float c_leaf = nt <= RT_WIDE_MAX_LEAF_TRIS ? area * (float)nt * A.cost_tri : INF;
    const float c_internal = dist[8] + area * A.cost_node;
    if (c_leaf <= c_internal)
        dec |= 1ull << 42;
    float c[7];
    int eff = 1;
    c[0] = fminf(c_leaf, c_internal);
    dec |= 1ull << 21;
#pragma unroll
    for (int i = 2; i <= 7; ++i) {
        if (dist[i] < c[i - 2]) {
            c[i - 1] = dist[i];
            eff = i;
        } else {
            c[i - 1] = c[i - 2];
        }
        dec |= (unsigned long long)eff << (21 + 3 * (i - 1));
    }
#pragma unroll
    for (int i = 0; i < 7; ++i)
        A.dp_cost[7ull * node + i] = c[i];
    A.dp_ntris[node] = nt;
    A.dp_dec[node] = dec;
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
            if (A.dp_cost)
                dp_node(A, node, lref, rref, l6, r6, box);
            child_is_leaf = 0u;
            child = node;
            node = A.node_parent[node];
        }
    }
}

// ---- 4b / 5b. PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) instead of the radix tree + refit: the
// Morton-ordered leaves are clusters; every round each cluster looks RADIUS positions to either side for the neighbour whose
// union box with it has the smallest surface area, mutual nearest neighbours merge into an inner node, the survivors are
// compacted (order kept), until one cluster is left. Agglomerative with a real surface-area criterion: the tree it builds is
// close to a full SAH build where the plain LBVH only ever splits at Morton-code bits. Boxes are unions of exact boxes, so
// every stored child box is still the exact bounding box of its subtree.
struct Ploc {
    uint32_t *ref[2];   // cluster -> leaf ref / inner node index, double buffered
    float *box[2];      // [6] per cluster
    uint32_t *nn;       // nearest neighbour (cluster position)
    uint32_t *valid;    // 1 = survives this round (also holds the merged cluster), 0 = absorbed
    uint32_t *pos;      // exclusive scan of valid
    uint32_t *depth;    // per inner node: height of its subtree (leaves 0)
    uint32_t *counters; // [0] inner nodes allocated, [1] root height (written with the last merge)
    uint32_t m;         // clusters this round
    int cur;            // which buffer holds them
    int radius;
    int force;          // no mutual pair last round (cannot happen with consistent tie-breaking): pair neighbours 2k, 2k + 1
};
#define PLOC_MAX_RADIUS 32
__global__ __launch_bounds__(256) void k_ploc_nn(const Ploc P) {
    __shared__ float s_box[256 + 2 * PLOC_MAX_RADIUS][6];
    const int m = (int)P.m, R = P.radius;
    const int block0 = (int)(blockIdx.x * blockDim.x);
    const float *box = P.box[P.cur];
    for (int t = (int)threadIdx.x; t < 256 + 2 * R; t += 256) {
        const int g = block0 - R + t;
#pragma unroll
        for (int c = 0; c < 6; ++c)
            s_box[t][c] = (g >= 0 && g < m) ? box[6ull * g + c] : 0.0f;
    }
    __syncthreads();
    const int i = block0 + (int)threadIdx.x;
    if (i >= m)
        return;
    if (P.force) {
        P.nn[i] = (uint32_t)((i ^ 1) < m ? (i ^ 1) : i);
        return;
    }
    const float *bi = s_box[threadIdx.x + R];
    float best = __builtin_inff();
    int bj = i;
    for (int dj = -R; dj <= R; ++dj) {
        const int j = i + dj;
        if (dj == 0 || j < 0 || j >= m)
            continue;
        const float *bjx = s_box[threadIdx.x + R + dj];
        float u[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            u[c] = fminf(bi[c], bjx[c]);
            u[3 + c] = fmaxf(bi[3 + c], bjx[3 + c]);
        }
        const float a = box_area6(u);
        // ascending j: on equal areas the lower position wins, on both sides of a pair — except that the position's pairing partner i ^ 1 wins every
        // tie it takes part in: when ALL areas tie (clusters collapsed onto a line or a point: zero-area unions) every position would otherwise choose
        // its lowest neighbour, only positions 0 and 1 would choose each other, and a round would merge ONE pair (found by tools/soak_wide_rays.py)
        if (a < best || (a == best && j == (i ^ 1))) {
            best = a;
            bj = j;
        }
    }
    P.nn[i] = (uint32_t)bj;
}
__global__ __launch_bounds__(256) void k_ploc_merge(const Ploc P, const BuildArrays A) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.m)
        return;
    const uint32_t j = P.nn[i];
    const bool mutual = j != i && P.nn[j] == i;
    if (!mutual) {
        P.valid[i] = 1u;
        return;
    }
    if (i > j) {
        P.valid[i] = 0u; // absorbed into position j
        return;
    }
    const uint32_t lref = P.ref[P.cur][i], rref = P.ref[P.cur][j];
    float l6[6], r6[6], u[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        l6[c] = P.box[P.cur][6ull * i + c];
        r6[c] = P.box[P.cur][6ull * j + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        u[c] = fminf(l6[c], r6[c]);
        u[3 + c] = fmaxf(l6[3 + c], r6[3 + c]);
    }
    const uint32_t node = atomicAdd(P.counters + 0, 1u);
    DevNode &nd = A.nodes[node];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        nd.lmin[c] = l6[c];
        nd.lmax[c] = l6[3 + c];
        nd.rmin[c] = r6[c];
        nd.rmax[c] = r6[3 + c];
    }
    nd.left = lref;
    nd.right = rref;
    nd.pad[0] = nd.pad[1] = 0;
    const uint32_t dl = (lref & RT_LEAF_FLAG) ? 0u : P.depth[lref], dr = (rref & RT_LEAF_FLAG) ? 0u : P.depth[rref];
    const uint32_t d = 1u + (dl > dr ? dl : dr);
    P.depth[node] = d;
    if (A.dp_cost)
        dp_node(A, node, lref, rref, l6, r6, u);
    // the merged cluster takes position i
    P.ref[P.cur][i] = node;
#pragma unroll
    for (int c = 0; c < 6; ++c)
        P.box[P.cur][6ull * i + c] = u[c];
    P.valid[i] = 1u;
    if (P.m == 2u)
        P.counters[1] = d; // the last merge: height of the whole tree
}
__global__ __launch_bounds__(256) void k_ploc_compact(const Ploc P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.m || !P.valid[i])
        return;
    const uint32_t q = P.pos[i];
    P.ref[P.cur ^ 1][q] = P.ref[P.cur][i];
#pragma unroll
    for (int c = 0; c < 6; ++c)
        P.box[P.cur ^ 1][6ull * q + c] = P.box[P.cur][6ull * i + c];
}
__global__ __launch_bounds__(256) void k_ploc_init(const Ploc P, const BuildArrays A) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n_leaves)
        return;
    P.ref[0][i] = leaf_ref(A, i);
#pragma unroll
    for (int c = 0; c < 6; ++c)
        P.box[0][6ull * i + c] = A.leaf_box[6ull * i + c];
}

// ---- 6. wide collapse, top down (RT_BUILD_WIDE on the device tree): one thread per wide node of the current level.
// The node's children are the roots its binary subtree was cut into by the dynamic program (same decisions, same left-to-right
// order as wide_build.cpp's `collect`); slots by octant (greedy on centroid . corner direction), boxes quantised floor / ceil on
// the node's power-of-two grid; inner children get consecutive records (one atomic per node), leaf slots consecutive triangle
// records (one atomic per node) copied from the Morton-ordered DevTri / DevAttr arrays of step 3.
struct WideEmit {
    const DevNode *nodes;               // the binary tree (inner nodes)
    const float *leaf_box;              // [n_leaves][6]
    const unsigned long long *dp_dec;
    const DevTri *tris_in;              // Morton order, one triangle per binary leaf
    const DevAttr *attrs_in;
    WideNode *wide;
    DevTri *tris_out;
    DevAttr *attrs_out;
    const uint2 *queue_in;              // {binary inner node, wide record index}
    uint2 *queue_out;
    uint32_t n_in;
    uint32_t *counters;                 // [0] wide records allocated, [1] triangle records allocated, [2] entries of queue_out
    WideGrid grid;                      // origins are snapped to it, cell exponents clamped to its 4-bit range (wide_grid.h), as in wide_build.cpp
};
__device__ __forceinline__ uint32_t dec_ksplit(unsigned long long d, int j) { return (uint32_t)(d >> (3 * (j - 2))) & 7u; }
__device__ __forceinline__ uint32_t dec_eff(unsigned long long d, int i) { return (uint32_t)(d >> (21 + 3 * (i - 1))) & 7u; }
__device__ __forceinline__ bool dec_as_leaf(unsigned long long d) { return (d >> 42) & 1ull; }

__global__ __launch_bounds__(64) void k_wide_emit(const WideEmit E) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= E.n_in)
        return;
    const uint32_t bnode = E.queue_in[q].x, widx = E.queue_in[q].y;
    // ---- the roots this node's subtree is cut into
    uint32_t kid[8];   // binary ref: inner index, or RT_LEAF_FLAG | ... for a single-triangle leaf
    bool kid_leaf[8];  // becomes a leaf slot (a binary leaf, or an inner node of <= 3 triangles the program chose to keep whole)
    float kbox[8][6];
    int nk = 0;
    {
        uint32_t st_ref[16];
        uint32_t st_i[16];
        float st_box[16][6];
        int sp = 0;
        const DevNode nd = E.nodes[bnode];
        const uint32_t k = dec_ksplit(E.dp_dec[bnode], 8);
        st_ref[sp] = nd.right, st_i[sp] = 8u - k;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            st_box[sp][c] = nd.rmin[c], st_box[sp][3 + c] = nd.rmax[c];
        ++sp;
        st_ref[sp] = nd.left, st_i[sp] = k;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            st_box[sp][c] = nd.lmin[c], st_box[sp][3 + c] = nd.lmax[c];
        ++sp;
        while (sp > 0) {
            --sp;
            const uint32_t m = st_ref[sp], i = st_i[sp];
            float b[6];
#pragma unroll
            for (int c = 0; c < 6; ++c)
                b[c] = st_box[sp][c];
            bool root_here = (m & RT_LEAF_FLAG) != 0u;
            unsigned long long d = 0ull;
            uint32_t j = 1u;
            if (!root_here) {
                d = E.dp_dec[m];
                j = dec_eff(d, (int)i);
                root_here = j == 1u;
            }
            if (root_here) {
                kid[nk] = m;
                kid_leaf[nk] = (m & RT_LEAF_FLAG) != 0u || dec_as_leaf(d);
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    kbox[nk][c] = b[c];
                ++nk;
                continue;
            }
            const DevNode mn = E.nodes[m];
            const uint32_t kk = dec_ksplit(d, (int)j);
            st_ref[sp] = mn.right, st_i[sp] = j - kk;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                st_box[sp][c] = mn.rmin[c], st_box[sp][3 + c] = mn.rmax[c];
            ++sp;
            st_ref[sp] = mn.left, st_i[sp] = kk;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                st_box[sp][c] = mn.lmin[c], st_box[sp][3 + c] = mn.lmax[c];
            ++sp;
        }
    }
    // ---- node box, grid
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = 0; i < nk; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], kbox[i][c]);
            hi[c] = fmaxf(hi[c], kbox[i][3 + c]);
        }
    WideNode rec;
    int ebias[3];
    float org[3]; // the node's origin: its lower corner snapped down to the scene's origin grid (what the quantised planes are measured from)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        rec.p[c] = org[c] = wide_snap_origin(E.grid, c, lo[c], nullptr);
        ebias[c] = wide_cell_exponent(E.grid, (double)hi[c] - (double)org[c]) + 127;
        rec.e[c] = (uint8_t)ebias[c];
    }
    // ---- slots: greedy on dot(child centre - node centre, corner direction of the slot)
    int child_in[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
        child_in[s] = -1;
    uint32_t assigned = 0u;
    for (int round = 0; round < nk; ++round) {
        float best = -__builtin_inff();
        int bi = -1, bs = -1;
        for (int i = 0; i < nk; ++i) {
            if (assigned & (1u << i))
                continue;
            float dc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                dc[c] = 0.5f * (kbox[i][c] + kbox[i][3 + c]) - 0.5f * (lo[c] + hi[c]);
            for (int s = 0; s < 8; ++s) {
                if (child_in[s] >= 0)
                    continue;
                const float sc = dc[0] * ((s & 1) ? 1.0f : -1.0f) + dc[1] * ((s & 2) ? 1.0f : -1.0f) + dc[2] * ((s & 4) ? 1.0f : -1.0f);
                if (sc > best) {
                    best = sc;
                    bi = i;
                    bs = s;
                }
            }
        }
        if (bi < 0) { // NaN boxes: any free pairing
            for (int i = 0; i < nk && bi < 0; ++i)
                if (!(assigned & (1u << i)))
                    bi = i;
            for (int s = 0; s < 8 && bs < 0; ++s)
                if (child_in[s] < 0)
                    bs = s;
        }
        child_in[bs] = bi;
        assigned |= 1u << bi;
    }
    // ---- counts, allocation
    uint32_t n_inner = 0, n_tri = 0;
    uint32_t cnt[8];
    for (int s = 0; s < 8; ++s) {
        cnt[s] = 0;
        const int i = child_in[s];
        if (i < 0)
            continue;
        if (!kid_leaf[i])
            ++n_inner;
        else {
            // triangles below: 1 for a binary leaf; an inner node kept whole has 2 or 3 single-triangle leaves below it
            uint32_t n = 1;
            if (!(kid[i] & RT_LEAF_FLAG)) {
                const DevNode a = E.nodes[kid[i]];
                n = 0;
                const uint32_t sub[2] = {a.left, a.right};
                for (int t = 0; t < 2; ++t) {
                    if (sub[t] & RT_LEAF_FLAG)
                        n += 1;
                    else
                        n += 2; // <= 3 triangles in all: an inner grandchild holds exactly two leaves
                }
            }
            cnt[s] = n;
            n_tri += n;
        }
    }
    const uint32_t first_child = n_inner ? atomicAdd(E.counters + 0, n_inner) : 0u;
    const uint32_t tri_base = n_tri ? atomicAdd(E.counters + 1, n_tri) : 0u;
    const uint32_t qbase = n_inner ? atomicAdd(E.counters + 2, n_inner) : 0u;
    rec.imask = 0;
    rec.child_base = first_child;
    rec.tri_base = tri_base;
    rec.tri_mask = 0;
    rec.pad = 0;
    uint32_t r_inner = 0, r_tri = 0;
    for (int s = 0; s < 8; ++s) {
        const int i = child_in[s];
        if (i < 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rec.qlo[c][s] = 255; // empty slot: inverted box
                rec.qhi[c][s] = 0;
            }
            continue;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double cell = ldexp(1.0, ebias[c] - 127);
            double ql = floor(((double)kbox[i][c] - (double)org[c]) / cell), qh = ceil(((double)kbox[i][3 + c] - (double)org[c]) / cell);
            ql = fmin(fmax(ql, 0.0), 255.0);
            qh = fmin(fmax(qh, 0.0), 255.0);
            while (ql > 0.0 && (double)org[c] + ql * cell > (double)kbox[i][c])
                ql -= 1.0;
            while (qh < 255.0 && (double)org[c] + qh * cell < (double)kbox[i][3 + c])
                qh += 1.0;
            rec.qlo[c][s] = (uint8_t)ql;
            rec.qhi[c][s] = (uint8_t)qh;
        }
        if (!kid_leaf[i]) {
            rec.imask |= (uint8_t)(1u << s);
            E.queue_out[qbase + r_inner] = make_uint2(kid[i], first_child + r_inner);
            ++r_inner;
        } else {
            // the slot's triangles, left to right
            uint32_t leaves[3];
            uint32_t nl = 0;
            if (kid[i] & RT_LEAF_FLAG) {
                leaves[nl++] = kid[i] & RT_LEAF_BEGIN_MASK;
            } else {
                const DevNode a = E.nodes[kid[i]];
                const uint32_t sub[2] = {a.left, a.right};
                for (int t = 0; t < 2; ++t) {
                    if (sub[t] & RT_LEAF_FLAG) {
                        leaves[nl++] = sub[t] & RT_LEAF_BEGIN_MASK;
                    } else {
                        const DevNode g = E.nodes[sub[t]];
                        leaves[nl++] = g.left & RT_LEAF_BEGIN_MASK;
                        leaves[nl++] = g.right & RT_LEAF_BEGIN_MASK;
                    }
                }
            }
            for (uint32_t t = 0; t < nl && t < RT_WIDE_MAX_LEAF_TRIS; ++t) {
                rec.tri_mask |= 1u << (3 * s + (int)t);
                DevTri tr = E.tris_in[leaves[t]];
                tr.flags = 0;
                E.tris_out[tri_base + r_tri] = tr;
                E.attrs_out[tri_base + r_tri] = E.attrs_in[leaves[t]];
                ++r_tri;
            }
        }
    }
    E.wide[widx] = rec;
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

hipError_t build_bvh_device(const rt_scene_desc *d, hipStream_t stream, DeviceBvh *out, const char **err, bool wide, float cost_node, float cost_tri) {
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
    if (d->build.lbvh_leaf_tris)
        leaf_tris = std::min(8u, std::max(1u, d->build.lbvh_leaf_tris));
    // the wide collapse regroups single-triangle leaves itself; scenes of a handful of triangles take the host collapse (rt_scene.cpp)
    wide = wide && n > 8u;
    if (wide)
        leaf_tris = 1;
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
    A.cost_node = cost_node, A.cost_tri = cost_tri;
    if (wide) {
        BUILD_TRY(tmp.alloc(&A.dp_cost, 7ull * n_leaves));
        BUILD_TRY(tmp.alloc(&A.dp_dec, (size_t)n_leaves));
        BUILD_TRY(tmp.alloc(&A.dp_ntris, (size_t)n_leaves));
    }
    BUILD_TRY(hipMemsetAsync(arrived, 0, 4ull * n_leaves, stream));
    BUILD_TRY(hipMemsetAsync(leaf_parent, 0xFF, 4ull * n_leaves, stream)); // RT_NONE: a single leaf has no parent
    const int lblocks = (int)std::min<uint64_t>(((uint64_t)n_leaves + 255) / 256, 256u * 16u);
    BUILD_TRY(RT_LAUNCH_CHECKED(k_leaves, dim3(lblocks), dim3(256), 0, stream, A));
    // binary tree over the leaves: PLOC (default) or the Karras radix tree + refit (rt_build_options.device_builder = RT_BUILDER_LBVH; also the fallback when
    // a PLOC tree comes out deeper than the traversal stacks allow)
    uint32_t bin_root = 0u;
    bool use_ploc = n_leaves > 1 && d->build.device_builder != RT_BUILDER_LBVH;
    if (use_ploc) {
        Ploc P{};
        // search radius: 8 positions to either side measured best on both bench scenes (S-sponza / S-10M, wide tree collapsed from
        // it: radius 2: 452 / 217 Msamples/s, 4: 495 / 224, 6: 502 / 224, 8: 496 / 244, 16: 464 / 228, 32: 465 / 231; profiles/r03_wide.txt)
        int radius = 8;
        if (d->build.ploc_radius)
            radius = std::min(PLOC_MAX_RADIUS, std::max(1, (int)d->build.ploc_radius));
        P.radius = radius;
        for (int k = 0; k < 2; ++k) {
            BUILD_TRY(tmp.alloc(&P.ref[k], (size_t)n_leaves));
            BUILD_TRY(tmp.alloc(&P.box[k], 6ull * n_leaves));
        }
        BUILD_TRY(tmp.alloc(&P.nn, (size_t)n_leaves));
        BUILD_TRY(tmp.alloc(&P.valid, (size_t)n_leaves));
        BUILD_TRY(tmp.alloc(&P.pos, (size_t)n_leaves));
        BUILD_TRY(tmp.alloc(&P.depth, (size_t)n_leaves));
        BUILD_TRY(tmp.alloc(&P.counters, (size_t)4));
        BUILD_TRY(hipMemsetAsync(P.counters, 0, 4 * sizeof(uint32_t), stream));
        size_t scan_bytes = 0;
        BUILD_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, P.valid, P.pos, 0u, (size_t)n_leaves, rocprim::plus<uint32_t>(), stream));
        char *scan_tmp;
        BUILD_TRY(tmp.alloc(&scan_tmp, scan_bytes));
        BUILD_TRY(RT_LAUNCH_CHECKED(k_ploc_init, dim3((n_leaves + 255u) / 256u), dim3(256), 0, stream, P, A)); // one thread per leaf (not grid-stride)
        P.m = n_leaves;
        P.cur = 0;
        int rounds = 0;
        while (P.m > 1) {
            const dim3 grid((P.m + 255u) / 256u);
            BUILD_TRY(RT_LAUNCH_CHECKED(k_ploc_nn, grid, dim3(256), 0, stream, P));
            BUILD_TRY(RT_LAUNCH_CHECKED(k_ploc_merge, grid, dim3(256), 0, stream, P, A));
            size_t sb = scan_bytes;
            BUILD_TRY(rocprim::exclusive_scan(scan_tmp, sb, P.valid, P.pos, 0u, (size_t)P.m, rocprim::plus<uint32_t>(), stream));
            BUILD_TRY(RT_LAUNCH_CHECKED(k_ploc_compact, grid, dim3(256), 0, stream, P));
            uint32_t last[2];
            BUILD_TRY(hipMemcpyAsync(&last[0], P.pos + (P.m - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            BUILD_TRY(hipMemcpyAsync(&last[1], P.valid + (P.m - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            BUILD_TRY(hipStreamSynchronize(stream));
            const uint32_t m_new = last[0] + last[1];
            P.force = (uint64_t)(P.m - m_new) * 32u < P.m ? 1 : 0; // (next to) no progress — a healthy round merges a quarter of the clusters —: pair neighbours next round
            P.m = m_new;
            P.cur ^= 1;
            if (++rounds > 4096) {
                if (err)
                    *err = "PLOC did not terminate";
                return hipErrorUnknown;
            }
        }
        uint32_t h_root, h_counters[4];
        BUILD_TRY(hipMemcpyAsync(&h_root, P.ref[P.cur], sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        BUILD_TRY(hipMemcpyAsync(h_counters, P.counters, sizeof(h_counters), hipMemcpyDeviceToHost, stream));
        BUILD_TRY(hipStreamSynchronize(stream));
        bin_root = h_root;
        out->rounds = (uint32_t)rounds;
        if (h_counters[0] != n_leaves - 1u || h_counters[1] > 60u) // deeper than the traversal stacks (RT_MAX_STACK 64): take the depth-bounded radix tree
            use_ploc = false;
    }
    if (n_leaves > 1 && !use_ploc) {
        BUILD_TRY(hipMemsetAsync(arrived, 0, 4ull * n_leaves, stream));
        BUILD_TRY(RT_LAUNCH_CHECKED(k_radix_tree, dim3(lblocks), dim3(256), 0, stream, A));
        BUILD_TRY(RT_LAUNCH_CHECKED(k_refit, dim3(lblocks), dim3(256), 0, stream, A));
        bin_root = 0u;
    }
    out->ploc = use_ploc;
    uint32_t h_bounds[8];
    BUILD_TRY(hipMemcpyAsync(h_bounds, bounds, sizeof(h_bounds), hipMemcpyDeviceToHost, stream));
    BUILD_TRY(hipStreamSynchronize(stream));
    const auto t2 = std::chrono::steady_clock::now();

    // ---- wide collapse on the device: level by level from the root; the host only reads each level's size
    WideNode *wide_nodes = nullptr;
    DevTri *wide_tris = nullptr;
    DevAttr *wide_attrs = nullptr;
    uint32_t n_wide = 0, wide_depth = 0;
    if (wide) {
        uint2 *queue[2];
        uint32_t *counters;
        BUILD_TRY(out_alloc((void **)&wide_nodes, sizeof(WideNode) * (size_t)n_leaves)); // <= one wide node per binary inner node
        BUILD_TRY(out_alloc((void **)&wide_tris, sizeof(DevTri) * (size_t)n));
        BUILD_TRY(out_alloc((void **)&wide_attrs, sizeof(DevAttr) * (size_t)n));
        BUILD_TRY(tmp.alloc(&queue[0], (size_t)n_leaves));
        BUILD_TRY(tmp.alloc(&queue[1], (size_t)n_leaves));
        BUILD_TRY(tmp.alloc(&counters, (size_t)4));
        const uint32_t init_counters[4] = {1u, 0u, 0u, 0u}; // record 0 = the root
        const uint2 root_entry = make_uint2(bin_root, 0u);   // the binary root (Karras: node 0; PLOC: the last merge) -> wide record 0
        BUILD_TRY(hipMemcpyAsync(counters, init_counters, sizeof(init_counters), hipMemcpyHostToDevice, stream));
        BUILD_TRY(hipMemcpyAsync(queue[0], &root_entry, sizeof(root_entry), hipMemcpyHostToDevice, stream));
        BUILD_TRY(hipStreamSynchronize(stream));
        WideEmit E{};
        {
            float s_lo[3], s_hi[3];
            for (int c = 0; c < 3; ++c)
                s_lo[c] = dec_f(h_bounds[c]), s_hi[c] = dec_f(h_bounds[3 + c]);
            E.grid = out->wide_grid = make_wide_grid(s_lo, s_hi);
        }
        E.nodes = A.nodes, E.leaf_box = A.leaf_box, E.dp_dec = A.dp_dec, E.tris_in = A.tris, E.attrs_in = A.attrs;
        E.wide = wide_nodes, E.tris_out = wide_tris, E.attrs_out = wide_attrs, E.counters = counters;
        uint32_t level_n = 1;
        int cur = 0;
        while (level_n > 0) {
            ++wide_depth;
            E.queue_in = queue[cur], E.queue_out = queue[cur ^ 1], E.n_in = level_n;
            BUILD_TRY(hipMemsetAsync(counters + 2, 0, sizeof(uint32_t), stream));
            BUILD_TRY(RT_LAUNCH_CHECKED(k_wide_emit, dim3((level_n + 63u) / 64u), dim3(64), 0, stream, E));
            uint32_t h_counters[4];
            BUILD_TRY(hipMemcpyAsync(h_counters, counters, sizeof(h_counters), hipMemcpyDeviceToHost, stream));
            BUILD_TRY(hipStreamSynchronize(stream));
            level_n = h_counters[2];
            n_wide = h_counters[0];
            if (wide_depth > 200u) { // cannot happen (a level always consumes its queue); never spin
                if (err)
                    *err = "wide collapse did not terminate";
                return hipErrorUnknown;
            }
            cur ^= 1;
        }
    }
    const auto t3 = std::chrono::steady_clock::now();

    guard.keep = true;
    out->wide = wide_nodes;
    out->n_wide = n_wide;
    out->wide_depth = wide_depth;
    out->wide_ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
    if (wide) { // the binary tree and the Morton-ordered records were scaffolding: only the wide tree stays
        (void)hipFree(A.nodes);
        (void)hipFree(A.tris);
        (void)hipFree(A.attrs);
        A.nodes = nullptr;
        A.tris = wide_tris;
        A.attrs = wide_attrs;
    }
    out->nodes = A.nodes;
    out->tris = A.tris;
    out->attrs = A.attrs;
    out->n_inner = n_leaves > 1 ? n_leaves - 1 : 0;
    out->n_tris = n;
    out->root = n_leaves > 1 ? bin_root : (RT_LEAF_FLAG | (n << 27) | 0u);
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

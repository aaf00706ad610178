// rt_wide_pack.hip — the 8-wide tree as the traversal kernels read it: 64-byte nodes and their triangle records in one array of 16-byte units
// (layout: rt_device_types.h, "what the wide kernels READ"). The builders (wide_build.cpp on the host, k_wide_emit on the device) emit 80-byte
// WideNode records with origins already snapped to the scene grid and exponents inside the 4-bit range (wide_grid.h); this pass is a lossless
// re-encoding plus a re-layout, on the device for either builder:
//   sizes      units(i) = 4 * inner children + 3 * leaf triangles of node i, rounded up to a whole node (64 B)
//   scan       base(i)  = 4 (the root's own record) + sum of units before i      -> groups in the order of the builder's node array
//   positions  child r of node i sits at base(i) + 4 r; the root at unit 0
//   write      header + planes at the node's position; its triangles behind its children, DevTri::pad = index into DevTri[] / DevAttr[]
// Why: an 80-byte record at an 80-byte stride costs a lane 5 vector-L1 accesses and crosses a 128-byte line half of the time; the packed node
// costs 4 and never does (profiles/r04_variants.txt: gather rates per record shape; the L1 access rate is what bounds wf_extend_wide).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "rt_kernels.h"
#include "rt_wide_pack.h"
#include "wide_grid.h"

namespace {

struct PackArgs {
    const WideNode *nodes;
    uint32_t n;
    const DevTri *tris;
    uint32_t *units;  // [n]
    uint32_t *base;   // [n] exclusive scan of units (without the root's 4)
    uint32_t *pos;    // [n] unit index of node i's own record
    uint4 *blob;
    WideGrid grid;
};

__global__ __launch_bounds__(256) void k_pack_sizes(const PackArgs A) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n)
        return;
    const uint4 h0 = reinterpret_cast<const uint4 *>(A.nodes + i)[0], h1 = reinterpret_cast<const uint4 *>(A.nodes + i)[1];
    const uint32_t n_inner = (uint32_t)__popc(h0.w >> 24), n_tri = (uint32_t)__popc(h1.z & 0xFFFFFFu);
    A.units[i] = (RT_WIDE_NODE_UNITS * n_inner + RT_WIDE_TRI_UNITS * n_tri + 3u) & ~3u;
    if (i == 0)
        A.pos[0] = 0u;
}
__global__ __launch_bounds__(256) void k_pack_positions(const PackArgs A) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n)
        return;
    const uint4 h0 = reinterpret_cast<const uint4 *>(A.nodes + i)[0], h1 = reinterpret_cast<const uint4 *>(A.nodes + i)[1];
    const uint32_t n_inner = (uint32_t)__popc(h0.w >> 24), child_base = h1.x;
    const uint32_t b = RT_WIDE_NODE_UNITS + A.base[i];
    for (uint32_t r = 0; r < n_inner; ++r)
        A.pos[child_base + r] = b + RT_WIDE_NODE_UNITS * r;
}
__global__ __launch_bounds__(256) void k_pack_write(const PackArgs A) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n)
        return;
    const uint4 *src = reinterpret_cast<const uint4 *>(A.nodes + i);
    const uint4 h0 = src[0], h1 = src[1];
    const uint32_t imask = h0.w >> 24, tri_mask = h1.z & 0xFFFFFFu, tri_base = h1.y;
    const uint32_t n_inner = (uint32_t)__popc(imask), n_tri = (uint32_t)__popc(tri_mask);
    const uint32_t b = RT_WIDE_NODE_UNITS + A.base[i];
    uint32_t state = tri_mask;
    for (uint32_t s = 0; s < 8u; ++s)
        if (imask & (1u << s))
            state |= 4u << (3u * s); // 100: an inner slot (it has no triangles: its three tri_mask bits are clear)
    uint32_t m[3], e4[3];
    const float p[3] = {__uint_as_float(h0.x), __uint_as_float(h0.y), __uint_as_float(h0.z)};
    for (int c = 0; c < 3; ++c) {
        (void)wide_snap_origin(A.grid, c, p[c], &m[c]); // p is on the grid already: this only recovers its index
        const int e = (int)((h0.w >> (8 * c)) & 255u) - 127 - A.grid.e_base;
        e4[c] = (uint32_t)(e < 0 ? 0 : (e > 15 ? 15 : e));
    }
    uint4 hdr;
    hdr.x = b;
    hdr.y = state | (e4[0] << 24) | (e4[1] << 28);
    hdr.z = m[0] | (m[1] << 20);
    hdr.w = (m[1] >> 12) | (m[2] << 8) | (e4[2] << 28);
    uint4 *dst = A.blob + A.pos[i];
    dst[0] = hdr;
    dst[1] = src[2];
    dst[2] = src[3];
    dst[3] = src[4];
    uint4 *tdst = A.blob + b + RT_WIDE_NODE_UNITS * n_inner;
    for (uint32_t j = 0; j < n_tri; ++j) {
        const uint4 *t = reinterpret_cast<const uint4 *>(A.tris + tri_base + j);
        uint4 r2 = t[2];
        r2.w = tri_base + j; // DevTri::pad: the record's index into DevTri[] / DevAttr[] (what a hit record stores)
        tdst[RT_WIDE_TRI_UNITS * j] = t[0];
        tdst[RT_WIDE_TRI_UNITS * j + 1] = t[1];
        tdst[RT_WIDE_TRI_UNITS * j + 2] = r2;
    }
}

} // namespace

namespace rt {

#define PACK_TRY(expr)                 \
    do {                               \
        hipError_t e_ = (expr);        \
        if (e_ != hipSuccess) {        \
            if (err)                   \
                *err = #expr;          \
            for (void *q_ : scratch)   \
                (void)hipFree(q_);     \
            if (blob)                  \
                (void)hipFree(blob);   \
            return e_;                 \
        }                              \
    } while (0)

hipError_t pack_wide_device(const WideNode *d_nodes, uint32_t n_wide, const DevTri *d_tris, const WideGrid &grid, hipStream_t stream, uint4_pod **blob_out,
                            uint32_t *n_units_out, const char **err) {
    *blob_out = nullptr;
    *n_units_out = 0;
    if (n_wide == 0)
        return hipSuccess;
    std::vector<void *> scratch;
    void *blob = nullptr;
    auto alloc = [&](void **p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes ? bytes : 16);
        if (e == hipSuccess)
            scratch.push_back(*p);
        return e;
    };
    PackArgs A{};
    A.nodes = d_nodes, A.n = n_wide, A.tris = d_tris, A.grid = grid;
    PACK_TRY(alloc((void **)&A.units, 4ull * n_wide));
    PACK_TRY(alloc((void **)&A.base, 4ull * n_wide));
    PACK_TRY(alloc((void **)&A.pos, 4ull * n_wide));
    const dim3 grid_dim((n_wide + 255u) / 256u), block(256);
    PACK_TRY(RT_LAUNCH_CHECKED(k_pack_sizes, grid_dim, block, 0, stream, A));
    size_t scan_bytes = 0;
    PACK_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, A.units, A.base, 0u, (size_t)n_wide, rocprim::plus<uint32_t>(), stream));
    void *scan_tmp = nullptr;
    PACK_TRY(alloc(&scan_tmp, scan_bytes));
    PACK_TRY(rocprim::exclusive_scan(scan_tmp, scan_bytes, A.units, A.base, 0u, (size_t)n_wide, rocprim::plus<uint32_t>(), stream));
    uint32_t last[2] = {0u, 0u};
    PACK_TRY(hipMemcpyAsync(&last[0], A.base + (n_wide - 1u), 4, hipMemcpyDeviceToHost, stream));
    PACK_TRY(hipMemcpyAsync(&last[1], A.units + (n_wide - 1u), 4, hipMemcpyDeviceToHost, stream));
    PACK_TRY(hipStreamSynchronize(stream));
    const uint64_t n_units = (uint64_t)RT_WIDE_NODE_UNITS + last[0] + last[1];
    if (n_units >= 0xFFFFFFF0ull) { // 64 GB of blob: beyond the 32-bit unit index
        if (err)
            *err = "wide blob exceeds 2^32 units";
        for (void *q : scratch)
            (void)hipFree(q);
        return hipErrorOutOfMemory;
    }
    PACK_TRY(hipMalloc(&blob, n_units * 16ull));
    A.blob = static_cast<uint4 *>(blob);
    PACK_TRY(hipMemsetAsync(blob, 0, n_units * 16ull, stream)); // the padding units between groups
    PACK_TRY(RT_LAUNCH_CHECKED(k_pack_positions, grid_dim, block, 0, stream, A));
    PACK_TRY(RT_LAUNCH_CHECKED(k_pack_write, grid_dim, block, 0, stream, A));
    PACK_TRY(hipStreamSynchronize(stream));
    for (void *q : scratch)
        (void)hipFree(q);
    *blob_out = static_cast<uint4_pod *>(blob);
    *n_units_out = (uint32_t)n_units;
    return hipSuccess;
}

} // namespace rt

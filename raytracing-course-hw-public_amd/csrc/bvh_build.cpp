// bvh_build.cpp — host SAH sweep builder that reproduces the reference's BVH topology exactly, then flattens it.
//
// Follows src/bvh.h:262-393: longest-axis choice (272-276), std::sort by centroid (278), prefix/suffix "surface
// area" sweep (280-297) using aabb::surface_area() = 2*dot(diag, diag.yxz()) (geometry.h:419-421; note this is
// 2*(dx*dy + dy*dx + dz*dz), kept as is), the score with its pref[i+1] index (302-310), leaf rules (343-346),
// depth cap 64 (371), pre-order node numbering (351-363).
//
// Identical topology needs (a) the same float expressions without contraction and (b) the same permutation out of
// std::sort for equal keys. (b) holds because introsort's decisions depend only on comparator outcomes and element
// positions: we sort {key, index} pairs with `a.key < b.key`, the reference sorts Object pointers with the same
// predicate on the same initial order, both with this toolchain's libstdc++.
//
// This builder stays on the host (SURVEY 8f-1 lists a device builder as a "next" row); it is not part of the timed
// render loop, exactly as the reference's BVH builds happen once before the pixel loop (raytracer.h:633).
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <thread>

namespace rt {
namespace {

inline float fmin_ref(float a, float b) { return (b < a) ? b : a; } // std::min(a, b)
inline float fmax_ref(float a, float b) { return (a < b) ? b : a; } // std::max(a, b)

struct Box {
    float lo[3] = {INFINITY, INFINITY, INFINITY};
    float hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    void extend_point(const float *p) { // aabb::extend(vec3) geometry.h:393-396
        for (int k = 0; k < 3; ++k) {
            lo[k] = fmin_ref(lo[k], p[k]);
            hi[k] = fmax_ref(hi[k], p[k]);
        }
    }
    void extend(const Box &b) { // geometry.h:398-401
        for (int k = 0; k < 3; ++k) {
            lo[k] = fmin_ref(lo[k], b.lo[k]);
            hi[k] = fmax_ref(hi[k], b.hi[k]);
        }
    }
    float surface_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2 * (dx * dy + dy * dx + dz * dz);
    }
};

struct Keyed {
    float key;
    uint32_t idx;
};

struct BuildData { // read-only while building
    std::vector<Box> tri_box;     // per original triangle
    std::vector<float> center[3]; // triangle::center() geometry.h:485-487
};

struct Builder { // one per worker thread: own node list and scratch, shared BuildData
    const BuildData &data;
    const std::vector<Box> &tri_box;
    const std::vector<float> (&center)[3];
    explicit Builder(const BuildData &d) : data(d), tri_box(d.tri_box), center(d.center) {}
    std::vector<HostNode> nodes;
    std::vector<float> pref, suf;
    std::vector<Keyed> scratch;
    std::vector<Box> sorted_box; // the boxes of the slice being split, in sorted order: ONE gather through objs[] per split, then
                                 // the two sweeps and the children's bounds read memory in order (the gathers were the build's bottleneck)

    Box bounds_of(const uint32_t *objs, size_t n) const { // bvh.h:315-321
        Box r;
        for (size_t i = 0; i < n; ++i)
            r.extend(tri_box[objs[i]]);
        return r;
    }
    // bounds_of(objs + first, n) right after split() sorted the slice: the same boxes in the same order, read from sorted_box
    Box bounds_of_sorted(size_t first, size_t n) const {
        Box r;
        for (size_t i = 0; i < n; ++i)
            r.extend(sorted_box[first + i]);
        return r;
    }

    size_t split(uint32_t *objs, size_t n, const Box &box) { // bvh.h:268-313
        float dx = box.hi[0] - box.lo[0], dy = box.hi[1] - box.lo[1], dz = box.hi[2] - box.lo[2];
        int axis = (dx >= dy && dx >= dz) ? 0 : (dy >= dz ? 1 : 2);
        const float *keys = center[axis].data();
        scratch.resize(n);
        for (size_t i = 0; i < n; ++i)
            scratch[i] = {keys[objs[i]], objs[i]};
        std::sort(scratch.begin(), scratch.end(), [](const Keyed &l, const Keyed &r) { return l.key < r.key; });
        sorted_box.resize(n);
        for (size_t i = 0; i < n; ++i) {
            objs[i] = scratch[i].idx;
            sorted_box[i] = tri_box[objs[i]];
        }

        pref.clear();
        suf.clear();
        Box acc;
        pref.push_back(acc.surface_area());
        for (size_t i = 0; i < n; ++i) {
            acc.extend(sorted_box[i]);
            pref.push_back(acc.surface_area());
        }
        acc = Box();
        suf.push_back(acc.surface_area());
        for (size_t i = n; i-- > 0;) {
            acc.extend(sorted_box[i]);
            suf.push_back(acc.surface_area());
        }
        size_t best = n; // objs.end(): "no split"
        float best_score = n * acc.surface_area();
        for (int i = 1; i < (int)n; ++i) {
            float score = i * pref[i + 1] + (n - i) * suf[n - i];
            if (score < best_score) {
                best_score = score;
                best = (size_t)i;
            }
        }
        return best;
    }

    uint32_t build(uint32_t offset, uint32_t *objs, size_t n, const Box &box, uint32_t min_node_size, uint32_t depth_left) { // bvh.h:323-366
        auto leaf = [&]() {
            HostNode nd;
            std::memcpy(nd.lo, box.lo, sizeof(nd.lo));
            std::memcpy(nd.hi, box.hi, sizeof(nd.hi));
            nd.left = nd.right = RT_NONE;
            nd.obj_begin = offset;
            nd.obj_end = (uint32_t)(offset + n);
            nodes.push_back(nd);
            return (uint32_t)(nodes.size() - 1);
        };
        if (depth_left == 0)
            return leaf();
        size_t mid = split(objs, n, box);
        size_t nl = mid, nr = n - mid;
        if (nl == 0 || nr == 0 || (nl < min_node_size && nr < min_node_size))
            return leaf();
        uint32_t idx = (uint32_t)nodes.size();
        HostNode nd;
        std::memcpy(nd.lo, box.lo, sizeof(nd.lo));
        std::memcpy(nd.hi, box.hi, sizeof(nd.hi));
        nd.left = nd.right = RT_NONE;
        nd.obj_begin = nd.obj_end = 0;
        nodes.push_back(nd);
        const Box lb = bounds_of_sorted(0, nl), rb = bounds_of_sorted(nl, nr); // before the recursion reuses sorted_box
        uint32_t l = build(offset, objs, nl, lb, min_node_size, depth_left - 1);
        uint32_t r = build((uint32_t)(offset + nl), objs + nl, nr, rb, min_node_size, depth_left - 1);
        nodes[idx].left = l;
        nodes[idx].right = r;
        return idx;
    }
};

} // namespace

// Subtrees are independent once a node has been split, so the top of the tree is built with one task per subtree
// (std::async) and the node lists are concatenated in the reference's pre-order numbering afterwards: parent, whole
// left subtree, whole right subtree (bvh.h:351-363). Every split still runs the sequential std::sort on its own slice, so
// topology and object order are bit-identical to the single-threaded (and the reference's) build.
static std::vector<HostNode> build_parallel(const BuildData &data, uint32_t offset, uint32_t *objs, size_t n, const Box &box, uint32_t depth_left,
                                            int par_levels) {
    Builder w(data);
    if (par_levels <= 0 || n < 65536) {
        w.build(offset, objs, n, box, 4, depth_left);
        return std::move(w.nodes);
    }
    HostNode nd;
    std::memcpy(nd.lo, box.lo, sizeof(nd.lo));
    std::memcpy(nd.hi, box.hi, sizeof(nd.hi));
    nd.left = nd.right = RT_NONE;
    nd.obj_begin = offset;
    nd.obj_end = (uint32_t)(offset + n);
    size_t mid = n;
    if (depth_left != 0)
        mid = w.split(objs, n, box);
    const size_t nl = mid, nr = n - mid;
    if (depth_left == 0 || nl == 0 || nr == 0 || (nl < 4 && nr < 4)) // leaf rules of bvh.h:336-346
        return {nd};
    nd.obj_begin = nd.obj_end = 0;
    const Box lb = w.bounds_of_sorted(0, nl), rb = w.bounds_of_sorted(nl, nr);
    { // this task only waits from here on: give its scratch back before the children allocate theirs
        std::vector<Keyed>().swap(w.scratch);
        std::vector<Box>().swap(w.sorted_box);
        std::vector<float>().swap(w.pref);
        std::vector<float>().swap(w.suf);
    }
    auto left_task = std::async(std::launch::async, build_parallel, std::cref(data), offset, objs, nl, lb, depth_left - 1, par_levels - 1);
    std::vector<HostNode> right = build_parallel(data, (uint32_t)(offset + nl), objs + nl, nr, rb, depth_left - 1, par_levels - 1);
    std::vector<HostNode> left = left_task.get();
    std::vector<HostNode> out;
    out.reserve(1 + left.size() + right.size());
    nd.left = 1;
    nd.right = (uint32_t)(1 + left.size());
    out.push_back(nd);
    auto append = [&out](const std::vector<HostNode> &blk) {
        const uint32_t shift = (uint32_t)out.size();
        for (HostNode c : blk) {
            if (c.left != RT_NONE)
                c.left += shift;
            if (c.right != RT_NONE)
                c.right += shift;
            out.push_back(c);
        }
    };
    append(left);
    append(right);
    return out;
}

HostBvh build_bvh(const float *positions, uint32_t n_total, const std::vector<uint32_t> &subset) {
    HostBvh out;
    if (n_total == 0) { // bvh.h:373-376
        out.root = RT_NONE;
        return out;
    }
    BuildData data;
    data.tri_box.resize(n_total);
    for (int k = 0; k < 3; ++k)
        data.center[k].resize(n_total);
    for (uint32_t t : subset) {
        const float *p = positions + 9 * (size_t)t;
        Box bx; // triangle::bounding_box geometry.h:489-495
        bx.extend_point(p);
        bx.extend_point(p + 3);
        bx.extend_point(p + 6);
        data.tri_box[t] = bx;
        for (int k = 0; k < 3; ++k)
            data.center[k][t] = (p[k] + p[3 + k] + p[6 + k]) / 3;
    }
    out.order = subset;
    Builder b(data);
    Box root_box = b.bounds_of(out.order.data(), out.order.size());
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    // Surface-area splits are far from balanced (a 10 : 90 cut is common), so a fixed number of parallel levels leaves most threads waiting
    // for the largest subtree. Every subtree of at least 65 536 triangles becomes a task instead (build_parallel's size test), however deep it
    // sits: a few hundred short-lived threads for 10^7 triangles, all cores busy until the end. One hardware thread: plain recursion.
    const int par_levels = hw > 1 ? 64 : 0;
    out.nodes = build_parallel(data, 0, out.order.data(), out.order.size(), root_box, 64, par_levels);
    out.root = 0;
    return out;
}

FlatBvh flatten_bvh(const HostBvh &bvh, const float *positions, uint32_t node_order) {
    FlatBvh f;
    if (bvh.root == RT_NONE)
        return f;
    // sorted triangle records
    f.tris.resize(bvh.order.size());
    for (size_t k = 0; k < bvh.order.size(); ++k) {
        const float *p = positions + 9 * (size_t)bvh.order[k];
        DevTri &t = f.tris[k];
        for (int c = 0; c < 3; ++c) {
            t.a[c] = p[c];
            t.v[c] = p[3 + c] - p[c];
            t.u[c] = p[6 + c] - p[c];
        }
        t.prim = bvh.order[k];
        t.flags = 0;
        t.pad = 0;
    }
    // inner nodes get device indices in pre-order among inner nodes (rt_build_options.node_order, development: 1 = breadth first,
    // 2 = sibling pairs: a node's two inner children adjacent, subtrees depth first)
    std::vector<uint32_t> dev_index(bvh.nodes.size(), RT_NONE);
    uint32_t n_inner = 0;
    auto is_inner = [&](uint32_t i) { return bvh.nodes[i].left != RT_NONE || bvh.nodes[i].right != RT_NONE; };
    for (size_t i = 0; i < bvh.nodes.size(); ++i) {
        const HostNode &nd = bvh.nodes[i];
        if (!is_inner((uint32_t)i) && nd.obj_end > nd.obj_begin) {
            f.tris[nd.obj_end - 1].flags |= 1u; // last triangle of its leaf
            f.tris[nd.obj_begin].flags |= 2u;   // first triangle of its leaf
        }
    }
    const int order_mode = (int)node_order;
    if (order_mode == 1 && is_inner(bvh.root)) {
        std::vector<uint32_t> q{bvh.root};
        for (size_t h = 0; h < q.size(); ++h) {
            dev_index[q[h]] = n_inner++;
            for (uint32_t c : {bvh.nodes[q[h]].left, bvh.nodes[q[h]].right})
                if (is_inner(c))
                    q.push_back(c);
        }
    } else if (order_mode == 2 && is_inner(bvh.root)) {
        dev_index[bvh.root] = n_inner++;
        std::vector<uint32_t> st{bvh.root};
        while (!st.empty()) {
            const uint32_t i = st.back();
            st.pop_back();
            const uint32_t l = bvh.nodes[i].left, r = bvh.nodes[i].right;
            if (is_inner(l))
                dev_index[l] = n_inner++;
            if (is_inner(r))
                dev_index[r] = n_inner++;
            if (is_inner(r))
                st.push_back(r);
            if (is_inner(l))
                st.push_back(l);
        }
    } else {
        for (size_t i = 0; i < bvh.nodes.size(); ++i)
            if (is_inner((uint32_t)i))
                dev_index[i] = n_inner++;
    }
    auto ref_of = [&](uint32_t node) -> uint32_t {
        const HostNode &nd = bvh.nodes[node];
        if (dev_index[node] != RT_NONE)
            return dev_index[node];
        const uint32_t cnt = nd.obj_end - nd.obj_begin;
        return RT_LEAF_FLAG | ((cnt >= 1 && cnt <= RT_LEAF_COOP_MAX) ? (cnt << 27) : 0u) | nd.obj_begin;
    };
    f.nodes.resize(n_inner);
    for (size_t i = 0; i < bvh.nodes.size(); ++i) {
        if (dev_index[i] == RT_NONE)
            continue;
        const HostNode &nd = bvh.nodes[i];
        DevNode &d = f.nodes[dev_index[i]];
        const HostNode &l = bvh.nodes[nd.left];
        const HostNode &r = bvh.nodes[nd.right];
        std::memcpy(d.lmin, l.lo, 12);
        std::memcpy(d.lmax, l.hi, 12);
        std::memcpy(d.rmin, r.lo, 12);
        std::memcpy(d.rmax, r.hi, 12);
        d.left = ref_of(nd.left);
        d.right = ref_of(nd.right);
        d.pad[0] = d.pad[1] = 0;
        for (const float *bx : {d.lmin, d.lmax, d.rmin, d.rmax})
            for (int c = 0; c < 3; ++c) {
                const float m = std::fabs(bx[c]);
                if (!(bx[c] == 0.0f || (m >= 7.275957614183426e-12f && m <= 1099511627776.0f)))
                    f.fast_ok = false;
            }
    }
    f.root = ref_of(bvh.root);
    return f;
}

} // namespace rt

extern "C" int rt_bvh_build_host(const float *positions, uint32_t n_triangles, const uint32_t *subset, uint32_t n_subset, uint32_t *n_nodes, uint32_t *root,
                                 uint32_t *nodes_out, uint32_t *order_out) {
    if ((n_triangles && !positions) || (n_subset && !subset) || !n_nodes || !root)
        return 1; // RT_ERR_INVALID_ARG
    std::vector<uint32_t> sub(subset, subset + n_subset);
    rt::HostBvh b = rt::build_bvh(positions, n_triangles, sub);
    *n_nodes = (uint32_t)b.nodes.size();
    *root = b.root;
    if (nodes_out)
        for (size_t i = 0; i < b.nodes.size(); ++i) {
            std::memcpy(nodes_out + 10 * i, b.nodes[i].lo, 12);
            std::memcpy(nodes_out + 10 * i + 3, b.nodes[i].hi, 12);
            nodes_out[10 * i + 6] = b.nodes[i].left;
            nodes_out[10 * i + 7] = b.nodes[i].right;
            nodes_out[10 * i + 8] = b.nodes[i].obj_begin;
            nodes_out[10 * i + 9] = b.nodes[i].obj_end;
        }
    if (order_out && !b.order.empty())
        std::memcpy(order_out, b.order.data(), b.order.size() * sizeof(uint32_t));
    return 0;
}

// wide_build.cpp — binary BVH -> 8-wide BVH with quantised child boxes (WideNode, rt_device_types.h): the production build mode.
//
// The reference traverses a binary tree of 40-byte nodes with one box each (src/bvh.h:157-163, 195-235). Its parity mode here
// keeps that topology (bvh_build.cpp). This file is the other mode: the same triangles under a tree the reference does not have —
// eight children per node, boxes quantised to 8 bits per plane — chosen so that a ray pays fewer, fatter node visits
// (DESIGN.md "production traversal"). Nothing here changes a triangle test, so a hit's (b, c, t) stay bit-exact; only which
// boxes are tested on the way differs, and every quantised box CONTAINS the exact box it stands for.
//
// Steps:
//   1. leaves of more than one triangle are split into single-triangle leaves (object median on the longest centroid axis),
//      so the collapse is free to regroup them;
//   2. bottom-up dynamic program over the binary tree (Ylitie, Karras, Laine 2017, sec. 3): C(n, i) = cheapest way to stand for
//      subtree n with at most i roots (i = 1..7), a root being a leaf of <= 3 triangles or a wide node; surface-area heuristic;
//   3. top-down emission: a wide node's children are the roots its subtree was cut into; children are dealt to the 8 slots so
//      that slot s lies towards corner s of the node (greedy assignment on centroid . corner direction), which is what lets the
//      kernel visit them front to back from the ray's direction signs alone; inner children and leaf triangles are laid out
//      consecutively per node; boxes are quantised with floor / ceil on the node's power-of-two grid and verified.
#include "wide_build.h"
#include "wide_grid.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

namespace rt {
namespace {

struct Box3 {
    float lo[3], hi[3];
};
inline void box_reset(Box3 &b) {
    for (int c = 0; c < 3; ++c) {
        b.lo[c] = INFINITY;
        b.hi[c] = -INFINITY;
    }
}
inline void box_grow(Box3 &b, const Box3 &o) {
    for (int c = 0; c < 3; ++c) {
        b.lo[c] = std::min(b.lo[c], o.lo[c]);
        b.hi[c] = std::max(b.hi[c], o.hi[c]);
    }
}
inline double box_area(const Box3 &b) {
    const double dx = (double)b.hi[0] - b.lo[0], dy = (double)b.hi[1] - b.lo[1], dz = (double)b.hi[2] - b.lo[2];
    if (!(dx >= 0 && dy >= 0 && dz >= 0))
        return 0.0;
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}
inline Box3 tri_box(const float *p) {
    Box3 b;
    box_reset(b);
    for (int v = 0; v < 3; ++v)
        for (int c = 0; c < 3; ++c) {
            b.lo[c] = std::min(b.lo[c], p[3 * v + c]);
            b.hi[c] = std::max(b.hi[c], p[3 * v + c]);
        }
    return b;
}

// working tree: single-triangle leaves
struct WNode {
    Box3 box;
    uint32_t left, right; // RT_NONE: leaf
    uint32_t tri;         // leaf: original triangle index
    uint32_t ntris;       // triangles below
};

struct Work {
    const float *pos;
    std::vector<WNode> n;
    uint32_t add_leaf(uint32_t tri) {
        WNode w;
        w.box = tri_box(pos + 9 * (size_t)tri);
        w.left = w.right = RT_NONE;
        w.tri = tri;
        w.ntris = 1;
        n.push_back(w);
        return (uint32_t)n.size() - 1;
    }
    // a leaf range of the input tree -> a small binary subtree over its triangles
    uint32_t split_leaf(uint32_t *tris, uint32_t count) {
        if (count == 1)
            return add_leaf(tris[0]);
        Box3 cb;
        box_reset(cb);
        auto centre = [&](uint32_t t, int c) { return (pos[9 * (size_t)t + c] + pos[9 * (size_t)t + 3 + c] + pos[9 * (size_t)t + 6 + c]) * (1.0f / 3.0f); };
        for (uint32_t i = 0; i < count; ++i)
            for (int c = 0; c < 3; ++c) {
                const float x = centre(tris[i], c);
                cb.lo[c] = std::min(cb.lo[c], x);
                cb.hi[c] = std::max(cb.hi[c], x);
            }
        int axis = 0;
        for (int c = 1; c < 3; ++c)
            if (cb.hi[c] - cb.lo[c] > cb.hi[axis] - cb.lo[axis])
                axis = c;
        std::stable_sort(tris, tris + count, [&](uint32_t a, uint32_t b) { return centre(a, axis) < centre(b, axis); });
        const uint32_t half = count / 2;
        const uint32_t l = split_leaf(tris, half), r = split_leaf(tris + half, count - half);
        WNode w;
        w.box = n[l].box;
        box_grow(w.box, n[r].box);
        w.left = l;
        w.right = r;
        w.tri = RT_NONE;
        w.ntris = count;
        n.push_back(w);
        return (uint32_t)n.size() - 1;
    }
};

constexpr float INF_COST = std::numeric_limits<float>::infinity();

} // namespace

BinBvh bin_from_host(const HostBvh &h) {
    BinBvh b;
    b.order = h.order;
    b.root = h.root;
    b.nodes.resize(h.nodes.size());
    for (size_t i = 0; i < h.nodes.size(); ++i) {
        const HostNode &s = h.nodes[i];
        BinNode &d = b.nodes[i];
        std::memcpy(d.lo, s.lo, 12);
        std::memcpy(d.hi, s.hi, 12);
        d.left = s.left;
        d.right = s.right;
        d.first = s.obj_begin;
        d.count = (s.left == RT_NONE && s.right == RT_NONE) ? s.obj_end - s.obj_begin : 0u;
    }
    return b;
}

BinBvh bin_from_device(const std::vector<DevNode> &nodes, const std::vector<DevTri> &tris, uint32_t root) {
    BinBvh b;
    b.order.resize(tris.size());
    for (size_t k = 0; k < tris.size(); ++k)
        b.order[k] = tris[k].prim;
    if (root == RT_NONE)
        return b;
    auto leaf_count = [&](uint32_t ref) {
        uint32_t cnt = RT_LEAF_CNT(ref);
        if (cnt == 0) { // big leaf: per-triangle last flag
            uint32_t k = ref & RT_LEAF_BEGIN_MASK;
            cnt = 1;
            while (k < tris.size() && !(tris[k].flags & 1u)) {
                ++k;
                ++cnt;
            }
        }
        return cnt;
    };
    b.nodes.resize(nodes.size());
    auto make_leaf = [&](uint32_t ref, const float *lo, const float *hi) {
        BinNode l{};
        std::memcpy(l.lo, lo, 12);
        std::memcpy(l.hi, hi, 12);
        l.left = l.right = RT_NONE;
        l.first = ref & RT_LEAF_BEGIN_MASK;
        l.count = leaf_count(ref);
        b.nodes.push_back(l);
        return (uint32_t)b.nodes.size() - 1;
    };
    if (root & RT_LEAF_FLAG) { // the whole scene is one leaf
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        b.nodes.clear();
        b.root = make_leaf(root, lo, hi); // build_wide recomputes boxes of split leaves from the triangles
        return b;
    }
    for (size_t i = 0; i < nodes.size(); ++i) {
        const DevNode &s = nodes[i];
        BinNode d{};
        for (int c = 0; c < 3; ++c) {
            d.lo[c] = std::min(s.lmin[c], s.rmin[c]);
            d.hi[c] = std::max(s.lmax[c], s.rmax[c]);
        }
        d.left = (s.left & RT_LEAF_FLAG) ? make_leaf(s.left, s.lmin, s.lmax) : s.left;
        d.right = (s.right & RT_LEAF_FLAG) ? make_leaf(s.right, s.rmin, s.rmax) : s.right;
        d.first = d.count = 0;
        b.nodes[i] = d;
    }
    b.root = root;
    return b;
}

WideBvh build_wide(const BinBvh &bin, const float *positions, float cost_node, float cost_tri) {
    WideBvh out;
    if (bin.root == RT_NONE || bin.order.empty())
        return out;
    // ---- 1. working tree with single-triangle leaves (iterative copy, parents after children is not required here)
    Work W;
    W.pos = positions;
    W.n.reserve(bin.order.size() * 2);
    std::vector<uint32_t> order = bin.order; // leaf ranges are permuted in place by split_leaf
    std::vector<uint32_t> map(bin.nodes.size(), RT_NONE);
    {
        // post-order over the input tree
        std::vector<std::pair<uint32_t, int>> st;
        st.push_back({bin.root, 0});
        while (!st.empty()) {
            auto [i, phase] = st.back();
            const BinNode &s = bin.nodes[i];
            if (s.left == RT_NONE && s.right == RT_NONE) {
                st.pop_back();
                map[i] = s.count ? W.split_leaf(order.data() + s.first, s.count) : RT_NONE;
                continue;
            }
            if (phase == 0) {
                st.back().second = 1;
                st.push_back({s.right, 0});
                st.push_back({s.left, 0});
                continue;
            }
            st.pop_back();
            const uint32_t l = map[s.left], r = map[s.right];
            if (l == RT_NONE || r == RT_NONE) { // an empty child (cannot happen in the builders here): pass the other one up
                map[i] = l == RT_NONE ? r : l;
                continue;
            }
            WNode w;
            w.box = W.n[l].box;
            box_grow(w.box, W.n[r].box);
            w.left = l;
            w.right = r;
            w.tri = RT_NONE;
            w.ntris = W.n[l].ntris + W.n[r].ntris;
            W.n.push_back(w);
            map[i] = (uint32_t)W.n.size() - 1;
        }
    }
    const uint32_t root = map[bin.root];
    if (root == RT_NONE)
        return out;
    const std::vector<WNode> &N = W.n; // children precede parents in index order: a plain loop is a bottom-up pass
    const size_t nn = N.size();
    const double root_area = std::max(box_area(N[root].box), 1e-300);
    out.grid = make_wide_grid(N[root].box.lo, N[root].box.hi);

    // ---- 2. the dynamic program
    std::vector<float> C(nn * 7);        // C[n*7 + i-1], i = 1..7
    std::vector<uint8_t> eff(nn * 7);    // roots actually used (<= i) for C(n, i); 1 = n itself is the root
    std::vector<uint8_t> ksplit(nn * 9); // ksplit[n*9 + j], j = 2..8: roots given to the left child when n is cut into j
    std::vector<uint8_t> as_leaf(nn);    // C(n, 1) was the leaf alternative
    for (size_t n = 0; n < nn; ++n) {
        const WNode &w = N[n];
        const float A = (float)(box_area(w.box) / root_area);
        const float c_leaf = w.ntris <= RT_WIDE_MAX_LEAF_TRIS ? A * (float)w.ntris * cost_tri : INF_COST;
        if (w.left == RT_NONE) {
            for (int i = 1; i <= 7; ++i) {
                C[n * 7 + i - 1] = c_leaf;
                eff[n * 7 + i - 1] = 1;
            }
            as_leaf[n] = 1;
            continue;
        }
        const float *CL = &C[(size_t)w.left * 7], *CR = &C[(size_t)w.right * 7];
        float dist[9];
        for (int j = 2; j <= 8; ++j) {
            float best = INF_COST;
            int bk = 1;
            for (int k = 1; k < j; ++k) {
                if (k > 7 || j - k > 7)
                    continue;
                const float c = CL[k - 1] + CR[j - k - 1];
                if (c < best) {
                    best = c;
                    bk = k;
                }
            }
            dist[j] = best;
            ksplit[n * 9 + j] = (uint8_t)bk;
        }
        const float c_internal = dist[8] + A * cost_node;
        as_leaf[n] = c_leaf <= c_internal;
        C[n * 7] = std::min(c_leaf, c_internal);
        eff[n * 7] = 1;
        for (int i = 2; i <= 7; ++i) {
            if (dist[i] < C[n * 7 + i - 2]) {
                C[n * 7 + i - 1] = dist[i];
                eff[n * 7 + i - 1] = (uint8_t)i;
            } else {
                C[n * 7 + i - 1] = C[n * 7 + i - 2];
                eff[n * 7 + i - 1] = eff[n * 7 + i - 2];
            }
        }
    }

    // ---- 3. emission
    struct Child {
        uint32_t node; // working-tree node
        bool leaf;
    };
    std::vector<Child> kids;
    // subtree m as at most i roots
    auto collect = [&](auto &&self, uint32_t m, int i) -> void {
        const int j = eff[(size_t)m * 7 + i - 1];
        if (j == 1) {
            kids.push_back({m, as_leaf[m] != 0});
            return;
        }
        const int k = ksplit[(size_t)m * 9 + j];
        self(self, N[m].left, k);
        self(self, N[m].right, j - k);
    };
    auto leaf_tris = [&](auto &&self, uint32_t m, std::vector<uint32_t> &dst) -> void {
        if (N[m].left == RT_NONE) {
            dst.push_back(N[m].tri);
            return;
        }
        self(self, N[m].left, dst);
        self(self, N[m].right, dst);
    };

    struct Pending {
        uint32_t wnode, index, depth;
    };
    std::vector<Pending> todo;
    out.nodes.emplace_back();
    todo.push_back({root, 0u, 1u});
    std::vector<uint32_t> tmp;
    while (!todo.empty()) {
        const Pending cur = todo.back();
        todo.pop_back();
        out.depth = std::max(out.depth, cur.depth);
        kids.clear();
        const WNode &w = N[cur.wnode];
        if (w.left == RT_NONE || (cur.wnode == root && as_leaf[root] && w.ntris <= RT_WIDE_MAX_LEAF_TRIS)) {
            kids.push_back({cur.wnode, true}); // a scene of <= 3 triangles: the root holds one leaf slot
        } else {
            const int k = ksplit[(size_t)cur.wnode * 9 + 8];
            collect(collect, w.left, k);
            collect(collect, w.right, 8 - k);
        }
        // slot assignment: greedy on dot(child centre - node centre, corner direction of the slot)
        const Box3 &nb = w.box;
        float ctr[3];
        for (int c = 0; c < 3; ++c)
            ctr[c] = 0.5f * (nb.lo[c] + nb.hi[c]);
        const int nk = (int)kids.size();
        float score[8][8];
        for (int i = 0; i < nk; ++i) {
            const Box3 &cb = N[kids[i].node].box;
            for (int s = 0; s < 8; ++s) {
                float d = 0;
                for (int c = 0; c < 3; ++c)
                    d += (0.5f * (cb.lo[c] + cb.hi[c]) - ctr[c]) * (((s >> c) & 1) ? 1.0f : -1.0f);
                score[i][s] = d;
            }
        }
        int slot_of[8], child_in[8];
        std::fill(slot_of, slot_of + 8, -1);
        std::fill(child_in, child_in + 8, -1);
        for (int round = 0; round < nk; ++round) {
            float best = -INFINITY;
            int bi = -1, bs = -1;
            for (int i = 0; i < nk; ++i)
                if (slot_of[i] < 0)
                    for (int s = 0; s < 8; ++s)
                        if (child_in[s] < 0 && score[i][s] > best) {
                            best = score[i][s];
                            bi = i;
                            bs = s;
                        }
            if (bi < 0) { // no score compares greater (NaN or -inf boxes of non-finite geometry): any free pairing, as the device collapse does
                for (int i = 0; i < nk && bi < 0; ++i)
                    if (slot_of[i] < 0)
                        bi = i;
                for (int s = 0; s < 8 && bs < 0; ++s)
                    if (child_in[s] < 0)
                        bs = s;
            }
            slot_of[bi] = bs;
            child_in[bs] = bi;
        }
        // the record
        // the record: origin snapped down to the scene's origin grid, cell exponents within the 4-bit range above e_base (wide_grid.h), so that
        // the pack pass (rt_wide_pack.hip) can re-encode the node in 64 bytes without touching a box
        WideNode rec;
        std::memset(&rec, 0, sizeof(rec));
        float org[3];
        for (int c = 0; c < 3; ++c)
            rec.p[c] = org[c] = wide_snap_origin(out.grid, c, nb.lo[c], nullptr);
        int ebias[3];
        double cell[3];
        for (int c = 0; c < 3; ++c) {
            const int e = wide_cell_exponent(out.grid, (double)nb.hi[c] - (double)org[c]);
            ebias[c] = e + 127;
            cell[c] = std::ldexp(1.0, e);
            rec.e[c] = (uint8_t)ebias[c];
        }
        for (int s = 0; s < 8; ++s)
            for (int c = 0; c < 3; ++c) {
                rec.qlo[c][s] = 255; // empty slot: inverted box
                rec.qhi[c][s] = 0;
            }
        rec.child_base = (uint32_t)out.nodes.size();
        rec.tri_base = (uint32_t)out.order.size();
        uint32_t n_inner = 0;
        for (int s = 0; s < 8; ++s) {
            const int i = child_in[s];
            if (i < 0)
                continue;
            const Box3 &cb = N[kids[i].node].box;
            for (int c = 0; c < 3; ++c) {
                double ql = std::floor(((double)cb.lo[c] - (double)org[c]) / cell[c]);
                double qh = std::ceil(((double)cb.hi[c] - (double)org[c]) / cell[c]);
                ql = std::min(std::max(ql, 0.0), 255.0);
                qh = std::min(std::max(qh, 0.0), 255.0);
                // containment in exact arithmetic (doubles hold these sums exactly enough; nudge if a rounding went the wrong way)
                while (ql > 0 && (double)org[c] + ql * cell[c] > (double)cb.lo[c])
                    ql -= 1;
                while (qh < 255 && (double)org[c] + qh * cell[c] < (double)cb.hi[c])
                    qh += 1;
                rec.qlo[c][s] = (uint8_t)ql;
                rec.qhi[c][s] = (uint8_t)qh;
            }
            if (kids[i].leaf) {
                tmp.clear();
                leaf_tris(leaf_tris, kids[i].node, tmp);
                for (size_t j = 0; j < tmp.size(); ++j) {
                    rec.tri_mask |= 1u << (3 * s + (int)j);
                    out.order.push_back(tmp[j]);
                }
            } else {
                rec.imask |= (uint8_t)(1u << s);
                ++n_inner;
            }
        }
        // inner children: consecutive records in slot order; processed depth-first so that a subtree stays close in memory
        const size_t first_child = out.nodes.size();
        out.nodes.resize(first_child + n_inner);
        out.nodes[cur.index] = rec;
        uint32_t r = 0;
        std::vector<Pending> mine;
        for (int s = 0; s < 8; ++s)
            if (rec.imask & (1u << s))
                mine.push_back({kids[child_in[s]].node, (uint32_t)(first_child + r++), cur.depth + 1});
        for (size_t k = mine.size(); k-- > 0;)
            todo.push_back(mine[k]);
    }
    out.sah_cost = C[(size_t)root * 7];
    return out;
}

} // namespace rt

// Host-only entry point (include/rt_host.h): the production build without a GPU, for tests of the tree itself.
extern "C" int rt_bvh_wide_build_host(const float *positions, uint32_t n_triangles, float cost_node, float cost_tri, uint32_t *n_nodes, uint32_t *depth,
                                      double *sah_cost, uint32_t *nodes80, uint32_t nodes_capacity, uint32_t *order_out) {
    if ((n_triangles && !positions) || !n_nodes)
        return 1; // RT_ERR_INVALID_ARG
    std::vector<uint32_t> all(n_triangles);
    std::iota(all.begin(), all.end(), 0u);
    const rt::HostBvh h = rt::build_bvh(positions, n_triangles, all);
    const rt::WideBvh w = rt::build_wide(rt::bin_from_host(h), positions, cost_node, cost_tri);
    *n_nodes = (uint32_t)w.nodes.size();
    if (depth)
        *depth = w.depth;
    if (sah_cost)
        *sah_cost = w.sah_cost;
    if (nodes80) {
        if (nodes_capacity < w.nodes.size())
            return 1;
        if (!w.nodes.empty()) // an empty scene has no node: memcpy's source must not be null even for 0 bytes
            std::memcpy(nodes80, w.nodes.data(), w.nodes.size() * sizeof(WideNode));
    }
    if (order_out && !w.order.empty())
        std::memcpy(order_out, w.order.data(), w.order.size() * sizeof(uint32_t));
    return 0;
}

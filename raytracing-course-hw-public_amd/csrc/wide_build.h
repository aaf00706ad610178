// wide_build.h — host side of the production build mode (RT_BUILD_WIDE): a binary BVH is collapsed into the 8-wide,
// quantised layout of rt_device_types.h (WideNode). Internal; the public switch is rt_scene_desc.build_flags.
#pragma once
#include <cstdint>
#include <vector>

#include "bvh_build.h"
#include "rt_device_types.h"

namespace rt {

// A binary BVH in the one form the collapse reads: inner nodes have two children, leaves a range of `order`.
struct BinNode {
    float lo[3], hi[3];
    uint32_t left, right; // RT_NONE for a leaf
    uint32_t first, count; // leaf: order[first .. first + count)
};
struct BinBvh {
    std::vector<BinNode> nodes;
    std::vector<uint32_t> order; // original triangle indices
    uint32_t root = RT_NONE;
};
BinBvh bin_from_host(const HostBvh &h);                                             // the reference-topology tree (bvh_build.cpp)
BinBvh bin_from_device(const std::vector<DevNode> &nodes, const std::vector<DevTri> &tris, uint32_t root); // an LBVH read back from HBM

struct WideBvh {
    std::vector<WideNode> nodes; // nodes[0] is the root (empty scene: no nodes)
    std::vector<uint32_t> order; // original triangle index of DevTri / DevAttr record k
    double sah_cost = 0;         // the collapse's cost estimate of the tree it chose (root surface area = 1)
    uint32_t depth = 0;
    WideGrid grid{};             // the origin / exponent grids every node was snapped to (wide_grid.h): what the pack pass encodes against
};
// Collapse by the surface-area-heuristic dynamic program of Ylitie et al. 2017 (sec. 3): every binary subtree gets the cheapest
// representation as a forest of at most i wide nodes / leaves (i = 1..7), leaves hold at most RT_WIDE_MAX_LEAF_TRIS triangles.
// `positions`: 9 floats per original triangle. cost_node / cost_tri: the model's price of one node visit and one triangle test.
WideBvh build_wide(const BinBvh &bin, const float *positions, float cost_node = 1.0f, float cost_tri = 0.3f);

} // namespace rt

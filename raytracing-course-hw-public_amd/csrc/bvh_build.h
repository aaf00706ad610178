// bvh_build.h — host BVH construction in the reference's topology + flattening to the device layout.
#pragma once
#include <cstdint>
#include <vector>

#include "rt_device_types.h"

namespace rt {

struct HostNode { // the reference's BVHNode (bvh.h:157-163), pre-order numbering (bvh.h:351-363)
    float lo[3], hi[3];
    uint32_t left, right, obj_begin, obj_end;
};

struct HostBvh {
    std::vector<HostNode> nodes;
    std::vector<uint32_t> order; // BVH::objects as original triangle indices (bvh.h:166)
    uint32_t root = RT_NONE;
};

// BVH::build(objs, pred, min_node_size = 4, max_depth = 64) (bvh.h:368-393) over the triangles `subset`
// (original indices, in scene order) of `positions` (9 floats per triangle). `n_total` is scene.objects.size():
// the reference returns a root-less BVH only when the WHOLE scene is empty (bvh.h:373-376).
HostBvh build_bvh(const float *positions, uint32_t n_total, const std::vector<uint32_t> &subset);

struct FlatBvh {
    std::vector<DevNode> nodes; // inner nodes only, pre-order among inner nodes
    std::vector<DevTri> tris;   // leaf order == HostBvh::order
    uint32_t root = RT_NONE;
    bool fast_ok = true; // all box coordinates are 0 or within [2^-37, 2^40] in magnitude
};
// `node_order` (rt_build_options.node_order): 0 pre-order among inner nodes, 1 breadth first, 2 sibling pairs (placement only)
FlatBvh flatten_bvh(const HostBvh &bvh, const float *positions, uint32_t node_order = 0);

} // namespace rt

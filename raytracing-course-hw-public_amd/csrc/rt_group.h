// rt_group.h — multi-GPU rendering inside the library (internal; the public boundary is include/rt_abi.h).
//
// Replaces the reference's thread pool over 256-pixel spans (src/raytracer.h:636-665: hardware_concurrency() threads
// pulling spans from one atomic) at node scale: one process, one host thread + one scene replica per GPU, the image
// split into interleaved pixel blocks (block b -> GPU b % G), and ONE exchange at the end: every GPU's blocks
// are gathered on GPU 0 with grouped ncclSend / ncclRecv over xGMI (RCCL), then handed to the caller.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>

#include "../../include/rt_abi.h"

namespace rt {

struct Group;
struct PreparedScene; // rt_scene.cpp: the host half of rt_create (BVH builds, texture pool, ...), shared by all replicas

// rt_scene.cpp: build the host half once / put one replica of it on a device
int prepare(const rt_scene_desc *desc, std::shared_ptr<const PreparedScene> *out);
int create_replica(const rt_scene_desc *desc, const std::shared_ptr<const PreparedScene> &prep, int device, rt_scene **out);

// `devices` = HIP ordinals (n_devices >= 1). Prepares the scene once on the host, puts one replica on every entry and creates one RCCL
// communicator over them (ncclCommInitAll). RT_ERR_COMM when RCCL is missing or refuses the device set.
int group_create(const rt_scene_desc *desc, const int *devices, int n_devices, Group **out);
void group_destroy(Group *g);
int group_size(const Group *g);
rt_scene *group_primary(Group *g); // replica on devices[0]: serves the probe entry points
// rt_render (fb != null) / rt_render_rgb8 (rgb8 != null) over all GPUs of the group; the result lands in the caller's
// buffer (host memory, or device memory of devices[0] with RT_FLAG_DEVICE_FB)
int group_render(Group *g, const rt_params *p, float *fb, uint8_t *rgb8, rt_stats *stats);

} // namespace rt

/*
 * rt_abi.h — C-ABI drop-in boundary for the per-pixel Monte Carlo render loop.
 *
 * The reference (firelion9/raytracing-course-hw-public) has no plugin/FFI layer;
 * the one seam where the hot path is entered is the call
 *
 *     run_raytracer(const Scene &scene, Image &image)      src/main.cpp:37 -> src/raytracer.h:629
 *
 * Everything below replaces exactly that call (and what it owns: the two BVH
 * builds of raytracer.h:440-447 and the per-pixel sample loop of raytracer.h:618-627).
 * Plain pointers + sizes only; no C++ or torch types cross this boundary.
 *
 * Reference inputs consumed at the seam and their POD restatement here:
 *   scene.objects[i].shape      (geometry.h:458-503, 3 x vec3)      -> rt_scene_desc.positions  (9 floats / triangle)
 *   scene.objects[i].attrs      (geometry.h:633-637)                -> normals (9), texcoords (6), tangents (9)
 *   scene.objects[i].material   (geometry.h:604-613, one per Object)-> material_ids[i] into rt_material_desc[]
 *   material.*_tex pointers     (geometry.h:610-613)                -> texture indices, RT_TEX_NONE = built-in
 *                                                                      WHITE_TEXTURE / NORMAL_UP (geometry.h:601-602)
 *   Texture::data (color4 float = stb u8 / 255.0f, geometry.h:590-595) -> RGBA8 texels (the /255.0f and the per-lookup
 *                                                                      pow(c, 2.2f) of geometry.h:525-527 are done
 *                                                                      by bit-exact 256-entry tables)
 *   scene.camera                (scene.h:60-72)                     -> rt_camera
 *   scene.bg_color, scene.bg    (scene.h:75,81; main.cpp:28-31)     -> bg_color, bg_texture (RT_TEX_NONE = the 1x1 white default of
 *                                                                      USE_ENV_MAP=false, config.h:37; else the environment map)
 *   scene.ray_depth, samples    (scene.h:76-77)                     -> rt_scene_desc.ray_depth, rt_params.samples
 * Reference output at the seam:
 *   image.set_pixel(p_idx, render_pixel(...)) (raytracer.h:658) which tone-maps at once (image.h:40-42,79-82)
 *                                                                   -> linear float3 framebuffer; tone-map/quantise is a
 *                                                                      pure per-pixel host function applied afterwards
 *                                                                      (rt_tonemap_rgb8, restating image.h:49-82).
 * Errors: the reference throws std::runtime_error (caught in main.cpp:46-49). Nothing is thrown across this
 * ABI: every entry point returns RT_OK or an error code and rt_last_error() holds the message.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 4u /* 2: rt_scene_desc carries analytic primitives (scene-txt front end); 3: bg_texture (environment map);
                             4: every tuning knob is a field (rt_build_options, rt_params.sort_mode / packet_mode / max_paths ...): the library
                                reads no environment variable; progress callback; packet census in rt_stats */
#define RT_TEX_NONE (-1)
#define RT_ALL_DEVICES (-1) /* rt_create: one scene replica on every visible GPU + an RCCL communicator over them */

/* error codes */
enum {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = 1,
    RT_ERR_NO_DEVICE = 2,   /* HIP runtime / GPU missing: the product never falls back to a CPU path */
    RT_ERR_HIP = 3,
    RT_ERR_OOM = 4,
    RT_ERR_IO = 5,
    RT_ERR_FORMAT = 6,
    RT_ERR_COMM = 7,
    RT_ERR_UNSUPPORTED = 8 /* a precondition of an optional fast path does not hold on this host; use the plain path */
};

/* RNG / transcendental policy of the sample loop.
 *   RT_RNG_DEVICE : counter-seeded xoshiro128++ stream per (pixel, sample) (include/rt_devspec.h). Scheduling
 *                   independent: any tiling / GPU count gives the same framebuffer. This is the production mode.
 *   RT_RNG_REFERENCE : the reference's stream: std::minstd_rand seeded with the 256-pixel span index
 *                   (raytracer.h:458,648; config.h:13) and the libstdc++-11 distribution algorithms,
 *                   one sequential stream per span. On the GPU one lane walks one span (parity mode,
 *                   not a performance mode): the image is the reference binary's, byte for byte.
 * Everything else is the reference's arithmetic in both modes, transcendentals included: sin / cos of the sampled
 * azimuths are glibc's sinf / cosf restated bit for bit (rt_devspec.h rt_sincos_libm, compared with libm on every
 * float of [0, 2 pi]); the modes differ in the random stream only.
 */
enum { RT_RNG_DEVICE = 0, RT_RNG_REFERENCE = 1 };

typedef struct rt_camera {
    float position[3];
    float right[3];
    float up[3];
    float forward[3];
    float fov_x; /* radians, as Camera::fov_x (scene.h:67) */
} rt_camera;

typedef struct rt_texture_desc {
    uint32_t width;
    uint32_t height;
    const uint8_t *rgba8; /* width*height*4, row-major, as stbi_load(..., 4) returns (geometry.h:586) */
} rt_texture_desc;

typedef struct rt_material_desc {
    float color[4];    /* material::color (geometry.h:605) */
    float emission[3]; /* material::emission */
    float roughness;
    float metallic;
    float ior;
    int32_t color_tex;    /* index into textures or RT_TEX_NONE */
    int32_t emissive_tex;
    int32_t metallic_roughness_tex;
    int32_t normal_tex;
} rt_material_desc;

/* Analytic primitives of the scene-txt front end (BASELINE configs 1-2; sample_data scene files: "ELLIPSOID rx ry rz", "PLANE nx ny nz"
 * with POSITION / ROTATION x y z w). The reference at HEAD renders triangles only (geometry.h:505) and keeps just an unused
 * intersect_ray_sphere (raytracer.h:61-77), so their semantics are DEFINED here (include/rt_primspec.h) and shared by the
 * HIP kernels and the CPU oracle: "parity unpinned" against the reference, bit-exact between the two. BOX and TRIANGLE
 * primitives are turned into triangles by the loader and take the (pinned) triangle path. */
enum { RT_PRIM_ELLIPSOID = 1, RT_PRIM_PLANE = 2 };
typedef struct rt_primitive_desc {
    uint32_t kind;        /* RT_PRIM_* */
    uint32_t material_id; /* into rt_scene_desc.materials; its textures are not sampled (scene-txt materials have none) */
    float param[3];       /* ELLIPSOID: semi-axes; PLANE: normal (need not be unit length) */
    float position[3];
    float rotation[4];    /* quaternion x y z w (order of sample_data and of geometry.h:154-156) */
} rt_primitive_desc;

/* Build-time tuning of rt_create. The reference's knobs are constexpr (config.h:7-47); these are per scene, and all-zero means
 * "the measured default" for every field, so a zero-initialised descriptor behaves as before. (Up to ABI 3 these were environment
 * variables read inside the library; now only the CLI, host/main.cpp, and bench.py translate environment into fields.) */
enum { RT_BUILDER_PLOC = 0, RT_BUILDER_LBVH = 1 };
typedef struct rt_build_options {
    uint32_t device_builder;  /* RT_BUILD_DEVICE_LBVH only: RT_BUILDER_PLOC (default; falls back to the radix tree when deeper than the
                                 traversal stacks) or RT_BUILDER_LBVH (Karras radix tree + refit) */
    uint32_t ploc_radius;     /* PLOC nearest-neighbour search radius, 1..32; 0 = 8 (profiles/r03_wide.txt) */
    uint32_t lbvh_leaf_tris;  /* triangles per leaf of the device-built binary tree, 1..8; 0 = 1 */
    uint32_t node_order;      /* order of the reference-topology tree's inner nodes in HBM: 0 pre-order (default), 1 breadth first,
                                 2 sibling pairs. Placement only: hits and counters do not depend on it */
    float wide_cost_node;     /* surface-area cost of a wide-node visit in the collapse; 0 = 1.0 */
    float wide_cost_tri;      /* ... of a triangle test; 0 = 0.3 */
    uint32_t wide_order;      /* order of the WideNode records in HBM: RT_WIDE_ORDER_* (placement only) */
    uint32_t reserved;        /* 0 */
} rt_build_options;
/* Placement of the 8-wide tree's records (csrc/wide_build.cpp, csrc/rt_bvh_device.hip). Children of one node are always
 * consecutive (the traversal finds child i at child_base + rank); what differs is where the groups of siblings go. */
enum {
    RT_WIDE_ORDER_DEFAULT = 0,  /* what measured best for the builder in use */
    RT_WIDE_ORDER_LEVEL = 1,    /* level by level (breadth first): the device builder's natural emission order */
    RT_WIDE_ORDER_DFS = 2,      /* sibling groups in depth-first order of their parents */
    RT_WIDE_ORDER_TREELET = 3   /* van-Emde-Boas-like: a node's children group, then recursively each child's subtree in blocks of
                                   three levels, so that a descent of three levels stays inside one contiguous region */
};

typedef struct rt_scene_desc {
    uint32_t abi_version; /* RT_ABI_VERSION */
    uint32_t n_triangles;
    const float *positions;       /* 9*n : a.xyz b.xyz c.xyz */
    const float *normals;         /* 9*n */
    const float *texcoords;       /* 6*n */
    const float *tangents;        /* 9*n */
    const uint32_t *material_ids; /* n */
    uint32_t n_materials;
    const rt_material_desc *materials;
    uint32_t n_textures;
    const rt_texture_desc *textures;
    rt_camera camera;
    float bg_color[3];
    uint32_t ray_depth; /* Scene::ray_depth; 8 for glTF (scene.h:186, config.h:17) */
    uint32_t n_primitives; /* analytic primitives, tested by brute force next to the BVH (at most RT_MAX_PRIMITIVES) */
    const rt_primitive_desc *primitives;
    uint32_t build_flags; /* RT_BUILD_* */
    int32_t bg_texture;   /* Scene::bg (scene.h:81; main.cpp:29-31, config.h:36-38 USE_ENV_MAP / ENV_MAP_PATH): index into `textures` of the
                             environment map that Scene::bg_at (scene.h:83-89) samples by direction, RT_TEX_NONE = the reference's default, the
                             1x1 WHITE_TEXTURE (a constant bg_color background). Loaders: rt_loaded_set_env_map (rt_host.h) */
    rt_build_options build; /* all-zero = defaults */
} rt_scene_desc;
/* How rt_create builds the scene BVH (BVH::build, bvh.h:262-393):
 *   RT_BUILD_REFERENCE (default): on the host, in the reference's exact topology (same SAH sweep, same std::sort
 *       permutation) -> event counters and tie-breaking equal the reference's: the parity mode.
 *   RT_BUILD_DEVICE_LBVH: on the GPU (Morton sort + Karras radix tree + refit, csrc/rt_bvh_device.hip), tens of milliseconds
 *       for 10^7 triangles instead of seconds. Same closest hits (identical t), but a different topology: ties between
 *       equal-t triangles may resolve differently and the counters differ. Production mode for big scenes; also selected by
 *       the CLI's RT_BVH_DEVICE=1. The light BVH (emissive triangles only) is always built on the host.
 *   RT_BUILD_WIDE (may be combined with either binary builder): the binary tree is collapsed into an 8-wide tree whose nodes
 *       hold eight child boxes quantised conservatively to 8 bits per plane (80 B per node), chosen by a surface-area
 *       dynamic program; the wavefront pipeline then walks THAT tree with global-best culling and octant-ordered slots
 *       (csrc/wide_build.cpp, csrc/rt_wide.hip). Production mode: the closest hit is the reference's (t bit for bit; another
 *       index only on exact ties) with far fewer memory accesses per ray; event counters count wide nodes. The megakernel /
 *       reference-RNG parity renders are refused on such a scene (RT_ERR_UNSUPPORTED). The CLI sets it for RT_BVH_WIDE=1.
 *   RT_BUILD_WIDE_HOST_COLLAPSE (development; with RT_BUILD_DEVICE_LBVH | RT_BUILD_WIDE): read the device-built binary tree back and
 *       collapse it on the host (wide_build.cpp) instead of on the device — the cross-check of the device collapse.
 *   RT_BUILD_LIGHTS_GLOBAL (development): never stage the light BVH in wf_shade's LDS.
 *   RT_BUILD_GROUP_COPY (tests; multi-GPU scenes): replace the RCCL exchange by peer copies, which lets a one-GPU box rehearse G > 1
 *       with repeated ordinals (RCCL refuses those). RT_BUILD_GROUP_SELF_EXCHANGE (tests): the first GPU's own blocks also travel
 *       through ncclSend / ncclRecv, which exercises the RCCL path with G = 1. */
enum { RT_BUILD_REFERENCE = 0, RT_BUILD_DEVICE_LBVH = 1, RT_BUILD_WIDE = 2, RT_BUILD_WIDE_HOST_COLLAPSE = 4, RT_BUILD_LIGHTS_GLOBAL = 8,
       RT_BUILD_GROUP_COPY = 16, RT_BUILD_GROUP_SELF_EXCHANGE = 32 };
#define RT_MAX_PRIMITIVES 4096u

/* Progress report of a render: called on the calling thread after every finished pass (pixel tile x sample range) of a single-GPU
 * scene, `done` of `total` passes; a multi-GPU scene reports once per GPU that finished, from that GPU's host thread (calls are
 * serialised). The reference prints "%d/%d     \r" per finished span (raytracer.h:647); the CLI does the same per pass under RT_VERBOSE. */
typedef void (*rt_progress_fn)(uint32_t done, uint32_t total, void *user);
/* Coherence sort of the rays of bounces >= 1 (wavefront pipeline; ordering never changes a result) */
enum {
    RT_SORT_AUTO = 0,             /* octant + cell + sub-cone where the tree does not fit the caches, none for a cache-resident wide tree */
    RT_SORT_OFF = 1,
    RT_SORT_CELL_OCTANT = 2,      /* 64^3 origin cell, direction octant (21 bits) */
    RT_SORT_COARSE_CELL_DIR = 3,  /* 16^3 cell, 9-bit direction code */
    RT_SORT_OCTANT_CELL = 4,      /* octant, 64^3 cell */
    RT_SORT_CELL_OCTANT_CONE = 5, /* cell, octant, sub-cone (24 bits) */
    RT_SORT_OCTANT_CELL_CONE = 6, /* octant, cell, sub-cone: what AUTO picks */
    RT_SORT_OCTANT_FINE_CELL_CONE = 7 /* octant, 128^3 cell, sub-cone (27 bits) */
};
/* Primary rays as 64-ray packets (wf_extend_packet / wf_extend_wide_packet) */
enum {
    RT_PACKET_AUTO = 0, /* from 16 (binary tree) / 4 (wide tree) samples per pixel and pass, until the kernel's own census of a pass
                           shows fewer than packet_min_lanes lanes served per trip for this image size / samples per pass */
    RT_PACKET_OFF = 1,
    RT_PACKET_ON = 2
};

typedef struct rt_params {
    uint32_t width;
    uint32_t height;
    uint32_t samples;  /* SPP */
    uint32_t rng_mode; /* RT_RNG_* */
    uint64_t seed;     /* RT_RNG_DEVICE only */
    /* Image sharding: this call renders pixel blocks  b  with  b % shard_count == shard_index,
     * where block b = row-major pixels [b*shard_block, (b+1)*shard_block). Pixels of other shards are left
     * untouched in fb_rgb. shard_count = 0 or 1 renders everything. shard_block must be a multiple of 256
     * (the reference span, config.h:13) in RT_RNG_REFERENCE mode. */
    uint32_t shard_index;
    uint32_t shard_count;
    uint32_t shard_block;
    uint32_t flags; /* RT_FLAG_* */
    /* ---- ABI 4: tuning (all-zero = the measured defaults) and progress */
    uint32_t sort_mode;       /* RT_SORT_* */
    uint32_t packet_mode;     /* RT_PACKET_* */
    float packet_min_lanes;   /* RT_PACKET_AUTO: lanes served per packet trip below which later passes use the per-lane kernel;
                                 0 = 33 (binary tree, profiles/r02_packet.txt) / 20 (wide tree, profiles/r03_wide.txt) */
    uint32_t reserved0;       /* 0 */
    uint64_t max_paths;       /* paths (pixel, sample) per pass of the wavefront pipeline; 0 = 128 M, capped by free device memory */
    rt_progress_fn progress;  /* may be NULL */
    void *progress_user;
} rt_params;

enum {
    RT_FLAG_NONE = 0,
    RT_FLAG_DEVICE_FB = 1, /* fb_rgb is a device pointer (HBM resident); no D2H copy. The library writes it on the scene's
                              own non-blocking stream and synchronises that stream before returning, so results are
                              complete on return; PRECONDITION: the buffer must be idle on entry (the caller has
                              synchronised whatever stream last touched it, e.g. its allocation's zero-fill) — the
                              library's stream is not ordered with any caller stream. */
    RT_FLAG_COUNTERS = 2,  /* run the instrumented kernel variant and fill the event counters of rt_stats
                              (slower; timing fields are filled whenever `stats` is non-NULL) */
    RT_FLAG_MEGAKERNEL = 4, /* RT_RNG_DEVICE only: use the persistent one-lane-per-pixel megakernel instead of the
                              wavefront pipeline (same image bit for bit; kept as a cross-check. RT_RNG_REFERENCE
                              always uses the megakernel: its RNG stream is sequential per 256-pixel span) */
    RT_FLAG_GLOBAL_BEST = 8 /* production traversal (wavefront pipeline only): BVH::intersect_ray prunes a far child only
                              against the NEAR subtree's local best (bvh.h:216-223); with this flag every box is culled
                              against the GLOBAL best hit so far, which visits a subset of the reference's nodes. The
                              closest hit is the same (t bit for bit) except where a triangle's t rounds below its own
                              box's entry distance, or on exact ties; the event counters differ. Off by default: the
                              parity mode reproduces the reference's order and counters. The CLI sets it for RT_TRAVERSAL=global. */
};

/* Per-render statistics (optional out-parameter). Counters are layout independent event counts in the
 * reference's terms: what the reference algorithm touches, not what caches absorb. */
typedef struct rt_stats {
    uint64_t samples;
    uint64_t casts;          /* closest-hit traversals (raytracer.h:540-553) */
    uint64_t nodes_visited;  /* BVH::intersect_ray invocations (bvh.h:195) */
    uint64_t box_tests;      /* intersect(ray, aabb) calls (bvh.h:137) */
    uint64_t tri_tests;      /* intersect(ray, triangle) calls (bvh.h:52) */
    uint64_t shaded_hits;    /* to_intersection_info on closest hits (bvh.h:176) */
    uint64_t light_queries;  /* bvh_mix_dist::pdf calls (raytracer.h:363) */
    uint64_t light_nodes;
    uint64_t light_box_tests;
    uint64_t light_tri_tests;
    uint64_t light_hits;
    uint64_t texel_fetches;  /* texels read by Texture::sample (geometry.h:559-568), 4 per non-1x1 lookup */
    double kernel_ms;        /* device time of the render kernel(s), HIP events on the launch stream */
    double total_ms;         /* wall time of rt_render */
    double dominant_ms;      /* summed device time of the dominant kernel's launches (wf_extend; the megakernel when
                                that path is used), one HIP event pair per launch */
    uint32_t dominant_launches;
    uint32_t packet_lanes_x100; /* last packet census read back: lanes served per packet trip x 100 (0: no packet pass ran, or none
                                   has been read back yet) */
    uint32_t passes;            /* passes (pixel tile x sample range) of this render */
    uint32_t packet_passes;     /* ... whose primary rays went through the packet kernel */
} rt_stats;

typedef struct rt_scene rt_scene; /* opaque: device-resident scene + both BVHs */

/* Replaces RaytracerStaticContext(scene) (raytracer.h:440-454): builds scene_bvh and light_bvh with the
 * reference's SAH sweep (bvh.h:268-393) on the host, flattens them into the HBM layouts of DESIGN.md and
 * uploads everything to `device` (HIP ordinal). Caller keeps ownership of every pointer in `desc`. */
int rt_create(const rt_scene_desc *desc, int device, rt_scene **out);
void rt_destroy(rt_scene *scene);

/* Multi-GPU scenes (replaces the reference's thread pool over spans, raytracer.h:636-665, at node scale).
 * rt_create(desc, RT_ALL_DEVICES, &s) — or rt_create_on(desc, devices, n, &s) for an explicit list of HIP ordinals —
 * builds one replica of the scene per GPU (one host thread each) and ONE RCCL communicator over them (ncclCommInitAll),
 * inside this call. rt_render / rt_render_rgb8 on such a scene split the image into interleaved pixel blocks
 * (block b -> GPU b % G; 8 image rows per block unless rt_params.shard_block says otherwise; rt_params.shard_count must
 * be 0 or 1), render them concurrently and gather every GPU's blocks on the first GPU with grouped ncclSend/ncclRecv
 * (rgb8 with the device film: 3 B/pixel), then deliver the whole image to the caller's buffer (host memory, or memory of
 * the first GPU with RT_FLAG_DEVICE_FB). The image is bit-identical to a single-GPU render (per-(pixel, sample) seeding).
 * The probe entry points (rt_cast_rays, rt_light_pdf, rt_bvh_info, rt_film_rgb8) run on the first GPU's replica.
 * Errors: RT_ERR_COMM when RCCL cannot be loaded, refuses the device set, or an exchange fails.
 * Tests: build_flags RT_BUILD_GROUP_COPY / RT_BUILD_GROUP_SELF_EXCHANGE (above). */
int rt_create_on(const rt_scene_desc *desc, const int *devices, int n_devices, rt_scene **out);
int rt_scene_device_count(const rt_scene *scene); /* GPUs rendering for this scene (1 for rt_create on one device) */

/* Replaces run_raytracer(scene, image) (raytracer.h:629-674). Blocking. fb_rgb: width*height*3 floats,
 * row-major, y down, linear radiance = render_pixel(ctx,x,y) (raytracer.h:618-627). ray_depth == 0 is a
 * silent no-op like raytracer.h:630-631. `stats` may be NULL. */
int rt_render(rt_scene *scene, const rt_params *params, float *fb_rgb, rt_stats *stats);

/* Closest-hit probe: cast `n` rays through the scene BVH exactly as cast_ray (raytracer.h:540-553) with
 * min_dst = EPS. rays: 6*n floats (origin, dir). Outputs per ray: prim (original triangle index; n_triangles + i for
 * analytic primitive i; 0xFFFFFFFF for a miss), and bct[3] = (b, c, t) of bvh.h:83-85 ((0, 0, t) for an analytic
 * primitive). Used by the parity tests for bit-exact hit indices. */
int rt_cast_rays(rt_scene *scene, const float *rays, uint32_t n, uint32_t *prim_out, float *bct_out);

/* The same probe through the RENDERER's closest-hit kernels (the wavefront pipeline's wf_extend / wf_extend_packet over a
 * queue made of `rays`), so that tests exercise the production kernels on arbitrary rays. mode: RT_CAST_*. `stats` (optional)
 * receives the event counters of the launch (casts, nodes_visited, box_tests, tri_tests) and its device time. */
enum {
    RT_CAST_PROBE = 0,         /* = rt_cast_rays: one lane per ray, the reference's recursion as an explicit stack */
    RT_CAST_EXTEND = 1,        /* wf_extend, reference traversal order (parity mode) */
    RT_CAST_EXTEND_GLOBAL = 2, /* wf_extend, global-best pruning (RT_FLAG_GLOBAL_BEST) */
    RT_CAST_PACKET = 3,        /* wf_extend_packet (64 consecutive rays = one packet), reference order */
    RT_CAST_PACKET_GLOBAL = 4  /* wf_extend_packet, global-best pruning */
};
int rt_cast_rays_ex(rt_scene *scene, const float *rays, uint32_t n, uint32_t mode, uint32_t *prim_out, float *bct_out, rt_stats *stats);

/* Intersection-info probe: the closest hit of each ray as rt_cast_rays reports it, plus the two normals to_intersection_info
 * (bvh.h:80-121) hands to shade(): `normal` (the geometric normal, flipped to face the ray: bvh.h:86-87, 118) and `shading_normal`
 * (smooth normal through the normal map, flipped likewise: bvh.h:94-108, 119). For an analytic primitive both are its unit normal
 * facing the ray (rt_primspec.h). 3 floats each per ray, zeros for a miss; either output may be NULL. Lets the tests pin normals
 * against an independent evaluation (tests/test_gpu_txt.py: float64 closed forms for ELLIPSOID / PLANE). */
int rt_surface_normals(rt_scene *scene, const float *rays, uint32_t n, uint32_t *prim_out, float *t_out, float *normal_out, float *shading_normal_out);

/* Light-pdf probe: bvh_mix_dist::pdf (raytracer.h:363-375) for n (origin, dir) pairs. */
int rt_light_pdf(rt_scene *scene, const float *rays, uint32_t n, float *pdf_out);

/* Background probe: Scene::bg_at (scene.h:83-89) for n directions (3 floats each, as the render loop passes ray.dir: unit length is
 * the caller's business) -> n x rgb. With bg_texture = RT_TEX_NONE every answer is bg_color. */
int rt_bg_at(rt_scene *scene, const float *dirs, uint32_t n, float *rgb_out);

/* BVH introspection for parity tests: which = 0 scene_bvh, 1 light_bvh. Nodes are reported in the
 * reference's own pre-order numbering (bvh.h:157-163, 351-363): 10 x u32-sized words per node
 * {min.xyz, max.xyz (float bits), left, right, obj_begin, obj_end}; order = the BVH's object permutation
 * (BVH::objects, bvh.h:166) as original triangle indices. Pass NULL buffers to query counts. */
int rt_bvh_info(rt_scene *scene, int which, uint32_t *n_nodes, uint32_t *n_objects, uint32_t *root,
                uint32_t *nodes_out /* 10*n_nodes */, uint32_t *order_out /* n_objects */);

/* The BVH exactly as the traversal kernels see it, copied back from HBM (tests: validates what rt_create uploaded or built
 * on the device, not a host copy). nodes64: n_inner records of 16 words {lmin.xyz, lmax.xyz, rmin.xyz, rmax.xyz, left,
 * right, pad, pad}; a child ref is an inner index, or 0x80000000 | count << 27 | first triangle for a leaf. tris48: n_tris
 * records of 12 words {a.xyz, (b-a).xyz, (c-a).xyz, original triangle index, flags, pad}. Pass NULL buffers for the counts. */
int rt_bvh_device_dump(rt_scene *scene, int which, uint32_t *n_inner, uint32_t *n_tris, uint32_t *root, uint32_t *nodes64, uint32_t *tris48);
/* The 8-wide scene BVH of a scene built with RT_BUILD_WIDE, copied back from HBM: nodes80 = n_nodes records of 20 words
 * (layout: WideNode, csrc/rt_device_types.h — origin xyz, exponents + inner mask, first inner child, first triangle, triangle
 * mask, pad, then qlo[3][8] and qhi[3][8] bytes); tris48 as in rt_bvh_device_dump, in the wide tree's order. NULL buffers: counts. */
int rt_bvh_wide_dump(rt_scene *scene, uint32_t *n_nodes, uint32_t *n_tris, uint32_t *depth, uint32_t *nodes80, uint32_t *tris48);
/* Wall time of the last rt_create's scene-BVH build in ms: {host build + flatten, 0} or {device build, upload of the raw arrays}. */
int rt_build_times(const rt_scene *scene, double *build_ms, double *upload_ms);
/* ... and of the wide collapse on top of it (RT_BUILD_WIDE; 0 otherwise): on the host (wide_build.cpp) after a host build, on the device after a device build. */
int rt_build_times_ex(const rt_scene *scene, double *build_ms, double *upload_ms, double *wide_ms);

/* Film (image.h:49-82): ACES -> gamma 1/2.2 -> x255 -> clamp -> round -> u8. Host function; n pixels. */
void rt_tonemap_rgb8(const float *rgb, size_t n_pixels, uint8_t *out_rgb8);

/* The same film on the device. rt_render_rgb8 = run_raytracer(scene, image) with the reference's own
 * output type: Image::set_pixel tone-maps each finished pixel at once (image.h:40-42), so the image the reference
 * holds after raytracer.h:629-674 is rgb8. Same parameters, sharding and flags as rt_render; `rgb8` receives
 * width*height*3 bytes (a device pointer with RT_FLAG_DEVICE_FB; only this shard's pixels are written), byte-identical
 * to rt_tonemap_rgb8 of the rt_render framebuffer. The gamma stage uses a threshold table derived from, and verified
 * against, the host libm's powf when the first call is made; if that verification fails the call returns an error
 * (no approximation is ever substituted). rt_film_rgb8 applies the device film to a caller-supplied host array. */
int rt_render_rgb8(rt_scene *scene, const rt_params *params, uint8_t *rgb8, rt_stats *stats);
int rt_film_rgb8(rt_scene *scene, const float *rgb, size_t n_pixels, uint8_t *out_rgb8);

const char *rt_last_error(void);
uint32_t rt_abi_version(void);
/* First 16 hex digits of the sha256 over the device-side sources this library was built from (csrc/device_sources.txt): lets a
 * caller (bench.py, __graft_entry__.build) check that the binary it measures is the tree it describes. */
const char *rt_source_stamp(void);
int rt_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */

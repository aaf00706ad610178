// rt_device_types.h — HBM data layout of the render loop (see DESIGN.md "Data layout in HBM").
//
// The reference walks `std::vector<BVHNode>` (40 B: a node's OWN box + links, bvh.h:157-163) and, per triangle,
// chases a `const Object*` into 208-byte AoS objects of which 36 bytes are used (geometry.h:639-643). Here:
//   * DevNode  (64 B, one per INNER node): BOTH children's boxes + child refs -> one 64-byte fetch per visited
//     inner node instead of two dependent 40-byte ones; leaves have no node record at all.
//   * DevTri   (48 B, in BVH leaf order): a, b-a, c-a (exactly the operands of bvh.h:40-43), the original
//     primitive index and an end-of-leaf flag; three aligned 16-byte loads, contiguous within a leaf.
//   * DevAttr  (128 B, same order): what to_intersection_info (bvh.h:80-121) reads on a shaded hit.
//   * DevMaterial (64 B) + texture table + RGBA8 texel pool + two 256-entry float tables.
// Every lane of a wavefront follows its own ray, so accesses are per-lane gathers: records are sized and aligned
// so that one gather touches the minimum number of 64/128-byte lines.
#pragma once
#include <stdint.h>

#include "../../include/rt_abi.h"

#define RT_NONE 0xFFFFFFFFu
#define RT_PRIM_FLAG 0x40000000u /* hit record: k = RT_PRIM_FLAG | analytic primitive index (a DevTri index is < 2^27) */
#define RT_LEAF_FLAG 0x80000000u
/* leaf ref = RT_LEAF_FLAG | cnt << 27 | first DevTri index (27 bits). cnt = number of triangles if 1..8, 0 = "walk the
   leaf with the per-triangle first/last flags" (larger leaves). */
#define RT_LEAF_BEGIN_MASK 0x07FFFFFFu
#define RT_LEAF_CNT(ref) (((ref) >> 27) & 15u)
#define RT_LEAF_COOP_MAX 8u
#define RT_MAX_STACK 64      /* bvh.h:371 max_depth = 64 -> at most 64 deferred siblings */
#define RT_MAX_RAY_DEPTH 32u /* reference uses 8 (config.h:17) */
#define RT_SPAN 256u         /* config.h:13 */

struct alignas(16) DevNode {
    float lmin[3], lmax[3]; // bounding box of the left child (bvh.h:208)
    float rmin[3], rmax[3]; // bounding box of the right child (bvh.h:212)
    uint32_t left, right;   // inner: index into DevNode[]; leaf: RT_LEAF_FLAG | first DevTri index
    uint32_t pad[2];
};
static_assert(sizeof(DevNode) == 64, "DevNode must be 64 bytes");

struct alignas(16) DevTri {
    float a[3];
    float v[3]; // b - a   (triangle::v geometry.h:473)
    float u[3]; // c - a   (triangle::u geometry.h:475)
    uint32_t prim;  // original triangle index (scene.objects position)
    uint32_t flags; // bit0: last triangle of its leaf, bit1: first triangle of its leaf
    uint32_t pad;
};
static_assert(sizeof(DevTri) == 48, "DevTri must be 48 bytes");

struct alignas(16) DevAttr {
    float n[9];  // attrs.normals
    float tg[9]; // attrs.tangents
    float uv[6]; // attrs.tex_coords
    float gn[3]; // base_normal() = norm(crs(v,u)) (geometry.h:648-650), precomputed with the same float ops
    uint32_t material;
    uint32_t pad[4];
};
static_assert(sizeof(DevAttr) == 128, "DevAttr must be 128 bytes");

struct alignas(16) uint4_pod { // a 16-byte unit of the wide blob (no HIP vector types in this header: host code includes it too)
    uint32_t x, y, z, w;
};

struct alignas(16) DevLightAux { // per light triangle (light-BVH order): triangle::normal(), triangle::square()
    float normal[3];
    float area;
};

struct alignas(16) DevMaterial {
    float color[4];
    float emission[3];
    float roughness;
    float metallic;
    float ior;
    int32_t color_tex, emissive_tex, mr_tex, normal_tex; // -1 = WHITE_TEXTURE / NORMAL_UP
    int32_t tex_set;       // >= 0: EVERY texture this material samples is a member of ONE interleaved view set (rt_scene.cpp): the view of
                           //       one of its members (tex_sample_set: addresses once, a 16-byte load per record). -1: sample slot by slot
    uint32_t tex_set_info; // bits 0..3: slots present (1 colour, 2 emissive, 4 metallic-roughness, 8 normal); bits 4..5: record dword of view `tex_set`
};
static_assert(sizeof(DevMaterial) == 64, "DevMaterial must be 64 bytes");

// A texture VIEW: what a material slot samples. Texels are stored in tiles of 2^tw_log x 2^th_log records so that a
// bilinear 2x2 footprint usually lies in one 128-B line, and the textures one material samples at the same (u, v)
// (colour, normal, metallic-roughness, emissive of equal size) are interleaved record by record (stride 4 dwords: one
// line serves all of a hit's lookups). rt_create builds the views (rt_scene.cpp build_texture_views); the values
// Texture::sample returns are unchanged, only the addresses differ.
struct alignas(16) DevTexture {
    uint32_t width, height;
    uint32_t offset; // pool dword of this view's texel (0,0)
    uint32_t count;  // width*height (== 1 -> Texture::sample's 1x1 fast path, geometry.h:548-550)
    uint32_t stride; // dwords between consecutive records: 1 (stand-alone) or 4 (interleaved set)
    uint32_t tiles_x; // tiles per row of tiles (width padded up to the tile width)
    uint32_t tw_log, th_log;
};
static_assert(sizeof(DevTexture) == 32, "DevTexture must be 32 bytes");

// ---- production traversal: the 8-wide BVH with quantised child boxes (RT_BUILD_WIDE; wide_build.cpp, rt_wide.hip)
// One 80-byte record = five 16-byte pieces holds EIGHT child boxes, each quantised conservatively (floor / ceil) to 8 bits per
// plane on a per-node grid: origin p, per-axis power-of-two cell size 2^(e - 127). A lane that tests eight boxes pays 5 vector-L1
// accesses; the binary DevNode pays 4 for two boxes, and the L1 access rate is wf_extend's roof (profiles/r02_l1_roof.txt).
// After Ylitie, Karras, Laine, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs" (HPG 2017), re-cut:
//   * child slot s sits towards the corner (s&1 ? +x : -x, s&2 ? +y : -y, s&4 ? +z : -z) of the node, so that a ray whose
//     direction signs are `oct` visits the hit slots in the order of decreasing (s ^ oct ^ 7): front to back without sorting;
//   * inner children (bit s of imask) are consecutive records from child_base, in slot order;
//   * the triangles of the node's leaf slots are consecutive DevTri records from tri_base: bit 3s + j of tri_mask says that leaf
//     slot s holds a j-th triangle (at most 3 per slot), and that triangle is record tri_base + popcount(tri_mask below the bit).
// An empty slot has an inverted box (qlo 255, qhi 0) that no ray hits.
#ifdef RT_WIDE_NODE_PAD128 /* experiment: one node per 128-byte line (never straddles); same fields */
struct alignas(128) WideNode {
#else
struct alignas(16) WideNode {
#endif
    float p[3];           // grid origin = the node box's lower corner
    uint8_t e[3];         // biased exponents: cell size on axis a = 2^(e[a] - 127)
    uint8_t imask;        // bit s: slot s is an inner node
    uint32_t child_base;  // first inner child (index into WideNode[])
    uint32_t tri_base;    // first triangle of this node's leaf slots (index into DevTri[] / DevAttr[])
    uint32_t tri_mask;    // 24 bits, see above
    uint32_t pad;
    uint8_t qlo[3][8];    // [axis][slot]
    uint8_t qhi[3][8];
};
#ifndef RT_WIDE_NODE_PAD128
static_assert(sizeof(WideNode) == 80, "WideNode must be 80 bytes");
#endif
#define RT_WIDE_MAX_LEAF_TRIS 3u

// ---- what the wide kernels READ: the same tree, re-encoded (round 4). An 80-byte record at an 80-byte stride crosses a 128-byte line for
// half of all nodes and costs a lane 5 vector-L1 tag accesses (one per 16-byte piece), and that access rate is what bounds wf_extend_wide
// (profiles/r04_variants.txt). The packed node is 64 bytes, 64-byte aligned: 4 accesses, never more than one line, two nodes per line.
//   piece 0  w0  base: 16-byte unit index (into the blob below) of this node's first inner child; its leaf triangles follow its inner children
//            w1  bits 0..23 slot states, 3 bits per slot: 000 empty, 100 inner node, 001 / 011 / 111 a leaf slot of 1 / 2 / 3 triangles (for the
//                leaf slots this IS WideNode::tri_mask); bits 24..27 / 28..31: cell exponents of x / y, relative to WideGrid::e_base
//            w2  bits 0..19 origin x, bits 20..31 the low 12 bits of origin y        (origins are multiples of WideGrid::g above WideGrid::base:
//            w3  bits 0..7 the high 8 bits of origin y, 8..27 origin z, 28..31 exponent z    the builders snap every node origin to that grid)
//   pieces 1..3  the 48 plane bytes exactly as WideNode::qlo / qhi.
// Nodes and triangle records live in ONE array of 16-byte units (the "blob"): [root][children of node A][triangles of A][pad to 64 B][children of B]...
// A node is 4 units, a triangle record (DevTri: 48 B) 3 units; DevTri::pad of a record in the blob holds its index into DevTri[] / DevAttr[]
// (the hit records and wf_shade keep using that index). One 32-bit base per node addresses 64 GB.
struct WideGrid {
    float base[3]; // per axis: a multiple of g at or below the scene's lower corner
    float g;       // origin granularity, a power of two: every node origin is base + m * g with m < 2^20, exactly representable
    int32_t e_base; // cell exponents are stored as e - e_base in 4 bits (e = unbiased exponent of the cell size); smaller cells are clamped up
    uint32_t pad_;
};
#define RT_WIDE_ORIGIN_BITS 20u
#define RT_WIDE_NODE_UNITS 4u
#define RT_WIDE_TRI_UNITS 3u

struct DevBvh {
    const DevNode *nodes;
    const DevTri *tris;
    uint32_t root;   // RT_NONE (no objects at all), inner index, or RT_LEAF_FLAG|0
    uint32_t n_tris; // BVH::objects.size()
    uint32_t fast_ok; // every node box coordinate is 0 or has magnitude in [2^-37, 2^40] (div_exact_fast precondition)
    uint32_t lds_inner; // light BVH only: 0, or 1 + number of inner nodes when nodes + triangles + aux fit RT_SHADE_LIGHTS_F4 (wf_shade stages them in LDS)
    const uint4_pod *wide; // scene BVH only: non-null = the scene was built wide (RT_BUILD_WIDE): the packed blob (above), root = unit 0;
                           // `nodes` is null, `tris` (and DevScene::attrs) are in the wide tree's triangle order
    uint32_t n_wide;       // wide nodes
    uint32_t n_units;      // 16-byte units of the blob
    WideGrid grid;
};
#define RT_SHADE_LIGHTS_F4 384 /* 6 KB of LDS in wf_shade: 4 pieces per inner node + 4 per light triangle (e.g. 31 nodes + 64 lights) */

struct DevScene {
    DevBvh scene;
    DevBvh lights;
    const DevAttr *attrs;          // scene-BVH order
    const DevLightAux *light_aux;  // light-BVH order
    const DevMaterial *materials;
    const DevTexture *textures;
    const uint32_t *texels;        // RGBA8, r in the low byte
    const float *lut_linear;       // k / 255.0f            (geometry.h:593-594)
    const float *lut_gamma;        // powf(k / 255.0f, 2.2f) (geometry.h:525-527, 616, 620)
    float cam_pos[3], cam_right[3], cam_up[3], cam_fwd[3];
    float bg[3];
    uint32_t ray_depth;
    float bounds_lo[3], bounds_inv[3]; // scene AABB: lo and 1/extent, for the ray-ordering key of the wavefront pipeline
    const rt_primitive_desc *prims; // analytic primitives of the scene-txt front end (include/rt_primspec.h), brute force
    uint32_t n_prims;
    uint32_t n_triangles;           // scene.objects.size(): rt_cast_rays reports analytic primitive i as n_triangles + i
    int32_t bg_tex;                 // Scene::bg (scene.h:81): view of the environment map in `textures`, -1 = the 1x1 WHITE_TEXTURE default
    uint32_t pad_bg;
};

struct DevStats { // device-side counters, see rt_stats in include/rt_abi.h
    unsigned long long samples, casts, nodes, box_tests, tri_tests, shaded, lq, lnodes, lbox, ltri, lhits, texels;
};

// ---- wavefront pipeline (rt_wavefront.hip): path = one (pixel, sample); queues hold live paths between bounces
// Queue record, one per live path (64 B = one HBM request): the ray (first 32 B), what wf_extend needs to start its traversal
// without arithmetic (next 16 B) and the path's RNG state (last 16 B). The state of a path travels WITH its ray through the queues, so wf_shade finds ray
// and state with one gather and writes both with the coalesced queue traffic instead of a random per-path record.
struct alignas(64) WfPath {
    float o[3];
    float dx;
    float dy, dz;
    uint32_t path;  // path id within the pass = index into fold / sample_out (& WF_ORDER_SLOT_MASK); bits 30-31: the sampler class of the path's
                    //   next shade() (next_shade_class), a scheduling hint for wf_shade
    uint32_t depth; // low 16 bits: remaining trace_ray budget (raytracer.h:596); high 16 bits: pending shade() frames
    float r[3];     // 1 / d, IEEE division, computed where the ray is made (wf_generate / wf_shade are latency bound; wf_extend,
    uint32_t fast;  //   the issue-bound kernel, just loads it) + the per-ray half of div_exact_fast's preconditions
    uint32_t s[4];  // xoshiro128++ state
};
static_assert(sizeof(WfPath) == 64, "WfPath must be 64 bytes");
#define WF_ORDER_SLOT_MASK 0x3FFFFFFFu /* WfLaunch::order: queue slot; max_paths <= 2^30 */
#define WF_ORDER_CLASS_SHIFT 30
struct alignas(16) WfHit { // 16 B: closest hit of the ray in the same queue slot
    uint32_t k;            // DevTri index (scene-BVH order) or RT_NONE
    float b, c, t;
};
struct alignas(16) RtF4 {
    float x, y, z, w;
};
struct alignas(16) WfFold { // 32 B: one pending shade() frame, `emission + inner * scale` (raytracer.h:588-590)
    float e[3], pad0;
    float s[3], pad1;
};
enum {
    WF_CNT_IN = 0, WF_CNT_TICKET = 2, WF_CNT_SLOTS = 3,
    WF_CNT_XCD = 32 /* 8 per-partition work tickets, one 128-byte line each */, WF_CNT_XCD_STRIDE = 32,
    WF_CNT_WORDS = WF_CNT_XCD + 8 * WF_CNT_XCD_STRIDE, // queue counters + tickets: cleared at the start of every pass, tickets again per bounce
    // behind them, never touched by the per-pass / per-bounce clears (round 3 kept these at words 32 and 64, i.e. INSIDE the ticket
    // lines: wf_advance's ticket reset wiped the census before the host could read it, and the packet policy never engaged)
    WF_CNT_CENSUS = WF_CNT_WORDS,       // 2 x u64: packet trips, lanes served (wf_extend_packet / wf_extend_wide_packet)
    WF_CNT_DIAG = WF_CNT_CENSUS + 8,    // 32 x u64: development census words (-DRT_DIAG builds)
    WF_CNT_ALLOC_WORDS = WF_CNT_DIAG + 64
};
static_assert(WF_CNT_XCD + 7 * WF_CNT_XCD_STRIDE < WF_CNT_WORDS && WF_CNT_CENSUS >= WF_CNT_WORDS && WF_CNT_DIAG >= WF_CNT_CENSUS + 4 && (WF_CNT_CENSUS % 2) == 0 && (WF_CNT_DIAG % 2) == 0,
              "census / diag words must not overlap the ticket lines and must be 8-byte aligned");
// wf_shade appends a bounce's survivors to WF_STRIPES sub-queues instead of one: a returning atomic on ONE address completes
// every ~13 ns chip-wide, and one per wave (64 rays) of a 30 M-ray bounce made that single counter the whole kernel's clock
// (6.3 ms of 6.3 ms; profiles/r02_shade_atomic.txt). Wave slot w of the input queue (positions 64w..64w+63) appends to
// sub-queue w % WF_STRIPES, whose region of paths_out is sized for all of its slots (nothing can overflow) and whose counter
// has a cache line to itself. The regions' fill levels become a dense prefix (run_start) in wf_advance; the next bounce's
// ray-order pass (wf_sort_keys) maps dense index -> physical slot, and wf_extend / wf_shade reach rays through `order` anyway.
#define WF_STRIPES 64u
#define WF_STRIPE_WORDS 32u /* one counter per 128-byte line */
#define WF_STRIPE_BUF_WORDS (WF_STRIPES * WF_STRIPE_WORDS + WF_STRIPES + 1u) /* counters, then run_start[WF_STRIPES + 1] */
// first slot of sub-queue k when the producing launch had n_slots wave slots: slots are dealt round-robin
#ifdef __HIPCC__
__host__ __device__
#endif
inline uint32_t wf_stripe_base(uint32_t k, uint32_t n_slots) {
    const uint32_t q = n_slots / WF_STRIPES, r = n_slots % WF_STRIPES;
    return 64u * (k * q + (k < r ? k : r));
}

struct WfLaunch {
    uint32_t width, height, samples; // image, total SPP
    uint32_t first_sample, pass_samples; // this pass renders samples [first_sample, first_sample + pass_samples)
    uint32_t first_pixel, pass_pixels;   // ... of local pixels [first_pixel, first_pixel + pass_pixels) of this shard
    uint32_t n_paths;                    // pass_pixels * pass_samples
    uint32_t shard_index, shard_count, shard_block;
    uint32_t ray_depth;
    uint64_t seed;
    float tan_x, tan_y;
    WfPath *paths_in, *paths_out; // this bounce's queue / the next one (compacted survivors)
    WfHit *hits;             // closest hit of the ray processed at queue POSITION q (position in `order` when sorted): wf_extend's
                             // waves take contiguous positions, so a line of hits is filled by one wave within one chunk and
                             // leaves L2 as a full line (stored at the ray's own slot, sorted rays scattered 16-B stores over
                             // the whole array: 9x HBM write amplification, profiles/r02_write_amp.txt)
    WfFold *fold;            // [ray_depth][n_paths]: pending shade() frames, level by level: wf_fold (one lane per path) reads each level coalesced
    RtF4 *sample_out;        // [n_paths]: sanitised radiance of each finished sample
    RtF4 *accum;             // [pass_pixels]: running per-pixel sum across sample passes (reference order)
    float *fb;               // width*height*3
    void *stack_overflow;    // wf_extend's evicted traversal-stack frames: [RT_MAX_STACK][stack_stride] records of 16 B
    uint32_t stack_stride;   // = threads of the wf_extend grid
    uint32_t *counters;      // WF_CNT_*: IN = rays of this bounce, TICKET = wf_extend's work ticket, SLOTS = wave slots of the launch
                             // that wrote paths_in (0: paths_in is dense, as wf_generate leaves it)
    uint32_t *stripes;       // WF_STRIPE_BUF_WORDS: the sub-queue fill counters of the running wf_shade, then run_start[] of paths_in
    void *diag;              // development census (-DRT_DIAG), 32 x u64, else unused
    const uint32_t *order;   // optional: position q processes queue slot order[q] & WF_ORDER_SLOT_MASK (coherence sort; extend AND shade); null = identity.
                             // Bits 30-31 carry the ray's NEXT sampler class (top bits of WfPath's path word) through the sort: wf_shade's lane assignment reads them
    uint32_t *sort_keys[2];  // sort workspace: keys / slot indices, double buffered
    uint32_t *sort_vals[2];
    void *sort_temp;
    uint32_t use_packet;     // this pass's primary rays go through wf_extend_packet (rt_scene.cpp decides: RT_WF_PACKET, samples per
                             // pixel, and what the kernel's own census said on an earlier pass)
    unsigned long long *packet_census; // [2] device: trips, lanes served (summed over the launch's waves)
    uint32_t global_best;    // 0: the reference's traversal order and pruning (parity mode); 1: prune against the global best
                             // (RT_FLAG_GLOBAL_BEST, production traversal: a subset of the reference's node visits)
    uint32_t order_classed;  // `order` came through the ray-order sort and carries the class bits; 0: identity-like order (no sort ran), slots only
    uint32_t sort_mode;      // 0 off, 1 cell+octant, 2 coarse cell + direction code, 3 octant+cell, 4 cell+octant+sub-cone (default) (RT_WF_SORT)
    size_t sort_temp_bytes;
    DevStats *stats;         // may be null
};

struct RenderLaunch {
    uint32_t width, height, samples, rng_mode;
    uint64_t seed;
    uint32_t shard_index, shard_count, shard_block; // normalised: count >= 1, block >= 1
    uint32_t n_items;      // work items of THIS shard (pixels in device mode, 256-pixel spans in reference mode)
    uint32_t items_per_block; // work items per shard block
    float tan_x, tan_y;    // tan(fov_x/2), tan(fov_y/2) hoisted from gen_ray (raytracer.h:531-535)
    float *fb;             // width*height*3, device
    uint32_t *counter;     // work-item ticket
    DevStats *stats;       // may be null
};

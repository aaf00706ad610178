/*
 * rt_host.h — host-side callers of the hot path, C-ABI. These stay plain C++ on the CPU (BASELINE north_star):
 * the glTF loader (reference src/scene.h:183-501), the film (src/image.h) and the PPM writer. They produce /
 * consume exactly the POD arrays of rt_abi.h, so that tests and the CLI feed the HIP path and the CPU oracle
 * from one loader.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_loaded_scene rt_loaded_scene; /* owns the arrays an rt_scene_desc points into */

/* parse_gltf_scene(path, aspect) (scene.h:183): glTF 2.0 with external .bin / image URIs, indexed triangle
 * lists (mode 4) and strips (mode 5), node TRS + matrix, perspective camera, metallic-roughness materials,
 * KHR_materials_emissive_strength. Quirks of the reference loader are kept (see DESIGN.md "loader quirks").
 * Also sets bg_color = ENV_MAP_INTENSITY (main.cpp:28) and ray_depth = DEFAULT_RAY_DEPTH (scene.h:186). */
int rt_gltf_load(const char *path, float aspect, rt_loaded_scene **out);
/* The scene-txt front end (BASELINE configs 1-2: sample_data scene files; no reference implementation at HEAD, grammar and
 * semantics in csrc/host/txt_loader.cpp): BOX and TRIANGLE primitives become triangles, ELLIPSOID and PLANE become
 * rt_primitive_desc entries, COLOR / EMISSION / METALLIC / DIELECTRIC / IOR become materials, legacy point / directional
 * lights are parsed and ignored. rt_scene_load picks the loader by file extension (".txt" -> scene-txt, else glTF): it is
 * what the CLI calls. rt_loaded_info returns what a scene-txt file says about the image (DIMENSIONS, SAMPLES; 0 for glTF)
 * and how many NEW_LIGHT blocks were ignored. */
int rt_txt_load(const char *path, rt_loaded_scene **out);
int rt_scene_load(const char *path, float aspect, rt_loaded_scene **out);
int rt_loaded_info(const rt_loaded_scene *s, uint32_t *width, uint32_t *height, uint32_t *samples, uint32_t *ignored_lights);
/* main.cpp:28-31 with USE_ENV_MAP = true (config.h:36-38): scene.bg = Texture::load_img(path) and bg_color = intensity. Decodes the
 * picture (PNG / JPEG / Radiance HDR, as rt_image_decode_file), appends it to the scene's textures and points desc.bg_texture at it.
 * The reference fixes path and intensity at compile time ("env.hdr", 1); the CLI takes them from RT_ENV_MAP / RT_ENV_MAP_INTENSITY. */
int rt_loaded_set_env_map(rt_loaded_scene *s, const char *image_path, float intensity);
/* scene.h:479-498 with ADD_LIGHT_TRIANGLE = true (config.h:40-47): appends an emissive triangle given in the camera's frame (rel: 3 x (right, up,
 * forward) coordinates; the reference's constants are {10, 0, -0.1}, {0, 10, -0.1}, {0, -10, -0.1} and intensity 10) with a default material.
 * The CLI takes it from RT_LIGHT_TRIANGLE=1 (+ RT_LIGHT_TRIANGLE_INTENSITY). */
int rt_loaded_add_light_triangle(rt_loaded_scene *s, const float rel[9], float intensity);
/* USE_TEXTURES = false (config.h:31-32): every Texture::sample returns the texture's first texel (geometry.h:547-574), which is what a 1x1 texture
 * does: the textures are cut down to their first texel. CLI: RT_USE_TEXTURES=0. */
int rt_loaded_disable_textures(rt_loaded_scene *s);
const rt_scene_desc *rt_loaded_desc(const rt_loaded_scene *s);
void rt_loaded_free(rt_loaded_scene *s);

/* Image::write (image.h:34-38): binary PPM "P6\n<w> <h>\n255\n" + rgb8. Creates parent directories like
 * main.cpp:40-41. */
int rt_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb8);

/* The gamma + quantise stage of the film (image.h:61-82) as the device film uses it: thr[k] (k = 1..255) is the smallest
 * ACES value whose level is >= k, thr[0] = 0; special = levels of {NaN, negative finite, -inf}. Built from the host
 * libm's powf and verified against it; returns RT_OK only if the verification held (see host/film.cpp). */
int rt_film_table(float thr[256], uint32_t special[3]);

/* BVH::build(objs, pred) (bvh.h:368-393) on the host, without a GPU: the same builder rt_create uses (reference topology,
 * subtrees built in parallel). `subset` = original indices of the triangles that pass the predicate, in scene order.
 * Output as rt_bvh_info: 10 words per node in the reference's pre-order numbering + the object permutation. The node
 * buffer must hold 2*n_subset + 1 nodes. */
int rt_bvh_build_host(const float *positions, uint32_t n_triangles, const uint32_t *subset, uint32_t n_subset, uint32_t *n_nodes, uint32_t *root,
                      uint32_t *nodes_out, uint32_t *order_out);

/* The production build (RT_BUILD_WIDE, include/rt_abi.h) on the host, no GPU: the reference-topology tree of all triangles collapsed
 * into the 8-wide quantised tree. nodes80: 20 words per node (WideNode, csrc/rt_device_types.h), root = node 0; order_out: original
 * triangle index of triangle record k. Pass nodes80 = NULL to query n_nodes. */
int rt_bvh_wide_build_host(const float *positions, uint32_t n_triangles, float cost_node, float cost_tri, uint32_t *n_nodes, uint32_t *depth,
                           double *sah_cost, uint32_t *nodes80, uint32_t nodes_capacity, uint32_t *order_out);

/* Texture::load_img (geometry.h:584-598: stbi_load with 4 channels forced) for the formats this loader reads, told apart by
 * their signatures: PNG (every colour type / bit depth / Adam7 / tRNS) and JPEG (baseline, extended-sequential and progressive
 * Huffman, 8 bit, 1 or 3 components, any sampling factors that divide the maximum, restart intervals). Both return the bytes
 * stb_image v2.30 returns (pinned by fixtures decoded with the reference's own stb build). Other formats stb_image reads
 * (BMP, TGA, GIF, PSD, PNM) are refused with RT_ERR_FORMAT and a message that names the format. Radiance HDR pictures (the
 * reference's default environment map is "env.hdr") come back as stb_image's 8-bit conversion of them (gamma 2.2, scale 1), which is what
 * the reference's stbi_load call receives. Caller frees with rt_free. */
int rt_image_decode_file(const char *path, uint32_t *w, uint32_t *h, uint8_t **rgba8);
int rt_hdr_decode_file(const char *path, uint32_t *w, uint32_t *h, uint8_t **rgba8);
int rt_png_decode_file(const char *path, uint32_t *w, uint32_t *h, uint8_t **rgba8);
int rt_jpeg_decode_file(const char *path, uint32_t *w, uint32_t *h, uint8_t **rgba8);
void rt_free(void *p);

#ifdef __cplusplus
}
#endif
#endif

// main.cpp — host driver behind the reference's CLI:  run.sh <scene.gltf | scene.txt> <W> <H> <SPP> <out.ppm>
// Restates src/main.cpp:16-49: parse 5 positional arguments, load the scene, render, tone-map, write the PPM.
// The only change of substance is line 37 of the reference: run_raytracer(scene, img) becomes
// rt_create + rt_render_rgb8 through the C ABI (include/rt_abi.h): render and film (image.h) on the device.
// Optional environment: RT_DEVICE (HIP ordinal; unset = every visible GPU: replicas + RCCL gather inside the library, the
// node-scale counterpart of the thread pool of raytracer.h:636-665), RT_RNG_MODE (device|reference), RT_SEED,
// RT_ENV_MAP (+ RT_ENV_MAP_INTENSITY): the environment map the reference enables at compile time (config.h:36-38, main.cpp:28-31);
// RT_LIGHT_TRIANGLE=1 (+ RT_LIGHT_TRIANGLE_INTENSITY): its extra light source in camera coordinates (config.h:40-47, scene.h:479-498).
// Tuning (the library itself reads NO environment variable since ABI 4; this file translates them into rt_scene_desc / rt_params
// fields): RT_BVH_DEVICE=1, RT_BVH_WIDE=1 (production builds), RT_TRAVERSAL=global, RT_WF_SORT=<0..5>, RT_WF_PACKET=<0|1>,
// RT_WF_MAX_PATHS=<n>, RT_DEVICE_BUILDER=lbvh, RT_PLOC_RADIUS=<n>. RT_VERBOSE: the reference's progress line "%d/%d     \r"
// (raytracer.h:647) per finished pass, and a timing line on stderr.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/rt_abi.h"
#include "../../../include/rt_host.h"

static void print_progress(uint32_t done, uint32_t total, void *) {
    std::printf("%d/%d     \r", (int)done, (int)total); // raytracer.h:647 (there: spans; here: passes)
    std::fflush(stdout);
}

static int die(const char *what) {
    std::cerr << what << ": " << rt_last_error() << std::endl; // main.cpp:46-49
    return EXIT_FAILURE;
}

int main(int argc, char **argv) {
    if (argc < 6) {
        std::cerr << "Too few arguments: expected 6, got " << argc - 1 << std::endl; // main.cpp:17-21 (sic)
        return EXIT_FAILURE;
    }
    unsigned width = std::strtol(argv[2], nullptr, 10);
    unsigned height = std::strtol(argv[3], nullptr, 10);
    unsigned samples = std::strtol(argv[4], nullptr, 10);
    if ((int)width <= 0 || (int)height <= 0) { // Image ctor image.h:25-29
        std::cerr << "Illegal image size" << (int)width << "x" << (int)height << std::endl;
        return EXIT_FAILURE;
    }
    rt_loaded_scene *loaded = nullptr;
    // *.txt: the scene-txt front end (BASELINE configs 1-2; no parser at the reference's HEAD), anything else: glTF (main.cpp:27)
    if (rt_scene_load(argv[1], static_cast<float>(width) / height, &loaded) != RT_OK)
        return die("load");
    if (const char *env_map = std::getenv("RT_ENV_MAP")) { // USE_ENV_MAP / ENV_MAP_PATH / ENV_MAP_INTENSITY, config.h:36-38
        const char *k = std::getenv("RT_ENV_MAP_INTENSITY");
        if (rt_loaded_set_env_map(loaded, env_map, k ? std::strtof(k, nullptr) : 1.0f) != RT_OK) {
            rt_loaded_free(loaded);
            return die("environment map");
        }
    }
    if (const char *ut = std::getenv("RT_USE_TEXTURES"); ut && std::atoi(ut) == 0) // USE_TEXTURES = false, config.h:31-32
        rt_loaded_disable_textures(loaded);
    if (const char *lt = std::getenv("RT_LIGHT_TRIANGLE"); lt && std::atoi(lt) != 0) { // ADD_LIGHT_TRIANGLE / LIGHT_TRIANGLE_* of config.h:40-47
        const float rel[9] = {10, 0, -0.1f, 0, 10, -0.1f, 0, -10, -0.1f};
        const char *k = std::getenv("RT_LIGHT_TRIANGLE_INTENSITY");
        if (rt_loaded_add_light_triangle(loaded, rel, k ? std::strtof(k, nullptr) : 10.0f) != RT_OK) {
            rt_loaded_free(loaded);
            return die("light triangle");
        }
    }
    const char *dev_env = std::getenv("RT_DEVICE");
    rt_scene *scene = nullptr;
    const int device = dev_env ? std::atoi(dev_env) : (rt_device_count() > 1 ? RT_ALL_DEVICES : 0);
    auto env_on = [](const char *name) {
        const char *v = std::getenv(name);
        return v && std::atoi(v) != 0;
    };
    rt_scene_desc desc = *rt_loaded_desc(loaded); // the arrays stay the loader's; only the build options change
    if (env_on("RT_BVH_DEVICE"))
        desc.build_flags |= RT_BUILD_DEVICE_LBVH;
    if (env_on("RT_BVH_WIDE"))
        desc.build_flags |= RT_BUILD_WIDE;
    if (const char *b = std::getenv("RT_DEVICE_BUILDER"); b && !std::strcmp(b, "lbvh"))
        desc.build.device_builder = RT_BUILDER_LBVH;
    if (const char *r = std::getenv("RT_PLOC_RADIUS"))
        desc.build.ploc_radius = (uint32_t)std::atoi(r);
    int crc = rt_create(&desc, device, &scene);
    if (crc == RT_ERR_COMM && device == RT_ALL_DEVICES) {
        // the multi-GPU group could not be formed (librccl missing, or it refuses this device set): one GPU still renders
        std::cerr << "rt_create: " << rt_last_error() << "; falling back to GPU 0 (set RT_DEVICE to choose another)" << std::endl;
        crc = rt_create(&desc, 0, &scene);
    }
    if (crc != RT_OK) {
        rt_loaded_free(loaded);
        return die("rt_create");
    }
    rt_params p{};
    p.width = width;
    p.height = height;
    p.samples = samples;
    const char *mode = std::getenv("RT_RNG_MODE");
    p.rng_mode = (mode && !std::strcmp(mode, "reference")) ? RT_RNG_REFERENCE : RT_RNG_DEVICE;
    const char *seed = std::getenv("RT_SEED");
    p.seed = seed ? std::strtoull(seed, nullptr, 0) : 0;
    if (const char *t = std::getenv("RT_TRAVERSAL"); t && !std::strcmp(t, "global"))
        p.flags |= RT_FLAG_GLOBAL_BEST;
    if (const char *v = std::getenv("RT_WF_SORT"))
        p.sort_mode = (uint32_t)std::atoi(v) + 1u; // RT_SORT_OFF = 1, then the keys in the order of RT_SORT_*
    if (const char *v = std::getenv("RT_WF_PACKET"))
        p.packet_mode = std::atoi(v) ? RT_PACKET_ON : RT_PACKET_OFF;
    if (const char *v = std::getenv("RT_WF_MAX_PATHS"))
        p.max_paths = std::strtoull(v, nullptr, 0);
    const bool verbose = std::getenv("RT_VERBOSE") != nullptr;
    if (verbose)
        p.progress = print_progress;
    // Image::set_pixel tone-maps as pixels finish (image.h:40-42): the film runs on the device too, unless the host
    // film is asked for (RT_FILM=host) or the device film declines (RT_ERR_UNSUPPORTED: libm failed its self-check)
    std::vector<uint8_t> rgb8((size_t)width * height * 3, 0);
    rt_stats st{};
    const char *film = std::getenv("RT_FILM");
    int rc = (film && !std::strcmp(film, "host")) ? RT_ERR_UNSUPPORTED : rt_render_rgb8(scene, &p, rgb8.data(), &st);
    if (rc == RT_ERR_UNSUPPORTED) {
        std::vector<float> fb((size_t)width * height * 3, 0.0f);
        rc = rt_render(scene, &p, fb.data(), &st);
        if (rc == RT_OK)
            rt_tonemap_rgb8(fb.data(), (size_t)width * height, rgb8.data());
    }
    rt_destroy(scene);
    rt_loaded_free(loaded);
    if (rc != RT_OK)
        return die("rt_render");
    if (rt_write_ppm(argv[5], width, height, rgb8.data()) != RT_OK)
        return die("write");
    if (verbose)
        std::fprintf(stderr, "samples=%llu kernel_ms=%.3f Msamples/s=%.3f\n", (unsigned long long)st.samples, st.kernel_ms,
                     st.kernel_ms > 0 ? st.samples / st.kernel_ms / 1e3 : 0.0);
    return EXIT_SUCCESS;
}

// loaded_scene.h — what a host loader (glTF: gltf_loader.cpp, scene-txt: txt_loader.cpp) hands to the C ABI: an
// rt_scene_desc plus the arrays it points into.
#pragma once
#include <cstdint>
#include <vector>

#include "../../../include/rt_host.h"

struct rt_loaded_scene {
    rt_scene_desc desc{};
    std::vector<float> positions, normals, texcoords, tangents;
    std::vector<uint32_t> material_ids;
    std::vector<rt_material_desc> materials;
    std::vector<rt_texture_desc> textures;
    std::vector<uint8_t *> texels;
    std::vector<rt_primitive_desc> primitives; // scene-txt only
    // scene-txt only: what the file itself says about the image (the CLI arguments win), and what was parsed but ignored
    uint32_t file_width = 0, file_height = 0, file_samples = 0, ignored_lights = 0, ignored_light_commands = 0;
    ~rt_loaded_scene() {
        for (auto *p : texels)
            rt_free(p);
    }
};

// film.cpp — host film: tone-map + quantise + PPM writer, and the shared error slot.
// Restates reference src/image.h:34-38 (write) and :49-82 (ACES -> gamma -> u8). Stays on the host: it needs
// glibc powf to be byte-identical with the reference, and it is a pure per-pixel function of the
// linear framebuffer the HIP path returns.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <filesystem>
#include <string>

#include "../../../include/rt_host.h"
#include "../rt_error.h"

namespace rt {
std::string &last_error() {
    thread_local std::string e;
    return e;
}
int fail(int code, const std::string &msg) {
    last_error() = msg;
    return code;
}
} // namespace rt

extern "C" const char *rt_last_error(void) { return rt::last_error().c_str(); }
extern "C" uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }

// image.h:51-59 aces_tonemap, :61-64 tone_mapping (pow(x, 1/GAMMA)), :66-82 discretize
extern "C" void rt_tonemap_rgb8(const float *rgb, size_t n_pixels, uint8_t *out_rgb8) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    const float inv_gamma = 1 / 2.2f;
    for (size_t i = 0; i < 3 * n_pixels; ++i) {
        const float x = rgb[i];
        const float aces = (x * (a * x + b)) / (x * (c * x + d) + e);
        const float mapped = std::pow(aces, inv_gamma) * 255;
        out_rgb8[i] = static_cast<uint8_t>(std::round(std::clamp(mapped, 0.0f, 255.0f)));
    }
}

extern "C" int rt_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb8) {
    if (!path || !rgb8 || width == 0 || height == 0)
        return rt::fail(RT_ERR_INVALID_ARG, "Illegal image size" + std::to_string(width) + "x" + std::to_string(height)); // image.h:26
    std::error_code ec;
    std::filesystem::path out_path(path);
    if (out_path.has_parent_path())
        std::filesystem::create_directories(out_path.parent_path(), ec); // main.cpp:41
    FILE *f = std::fopen(path, "wb");
    if (!f)
        return rt::fail(RT_ERR_IO, std::string("cannot open ") + path);
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    size_t n = (size_t)width * height * 3;
    size_t wr = std::fwrite(rgb8, 1, n, f);
    std::fclose(f);
    if (wr != n)
        return rt::fail(RT_ERR_IO, std::string("short write to ") + path);
    return RT_OK;
}

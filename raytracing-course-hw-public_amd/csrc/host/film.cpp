// film.cpp — host film: tone-map + quantise + PPM writer, and the shared error slot.
// Restates reference src/image.h:34-38 (write) and :49-82 (ACES -> gamma -> u8). The host film needs glibc powf to
// be byte-identical with the reference; the device film (rt_film.hip) reuses it through a verified
// threshold table built here (film_table).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <string>
#include <vector>

#include "../../../include/rt_host.h"
#include "../rt_error.h"
#include "../rt_film.h"

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

// image.h:61-64 tone_mapping (pow(x, 1/GAMMA)) + :66-82 discretize, applied to the ACES output
static inline uint8_t quantise_gamma(float aces) {
    const float inv_gamma = 1 / 2.2f;
    const float mapped = std::pow(aces, inv_gamma) * 255;
    return static_cast<uint8_t>(std::round(std::clamp(mapped, 0.0f, 255.0f)));
}

// image.h:51-59 aces_tonemap, then the gamma + quantise stage
extern "C" void rt_tonemap_rgb8(const float *rgb, size_t n_pixels, uint8_t *out_rgb8) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    for (size_t i = 0; i < 3 * n_pixels; ++i) {
        const float x = rgb[i];
        const float aces = (x * (a * x + b)) / (x * (c * x + d) + e);
        out_rgb8[i] = quantise_gamma(aces);
    }
}

// Device film support (rt_film.hip): the gamma + quantise stage as a table of 255 thresholds over the ACES value.
// q(y) = quantise_gamma(y) is non-decreasing in y >= 0 because glibc's powf is monotone there; thr[k] is the smallest
// float with q >= k, found by bisection over the float bit patterns and then VERIFIED (neighbourhood of every
// threshold + a pseudo-random sweep) against q itself. The device evaluates ACES with the same five IEEE operations
// and looks the level up, so device film == host film by construction wherever the verification holds; if it ever
// fails the table is reported unusable and the device film refuses to run. special[] = q(NaN), q(-1) (every negative
// finite y gives powf = NaN), q(-inf).
namespace rt {
bool film_table(float thr[256], uint32_t special[3]) {
    static bool built = false, ok = false;
    static float t_thr[256];
    static uint32_t t_special[3];
    if (!built) {
        built = true;
        auto f_of = [](uint32_t bits) {
            float f;
            std::memcpy(&f, &bits, 4);
            return f;
        };
        const uint32_t INF = 0x7F800000u;
        ok = quantise_gamma(0.0f) == 0 && quantise_gamma(f_of(INF)) == 255;
        t_thr[0] = 0.0f;
        std::vector<uint32_t> tb(256, 0);
        for (int k = 1; k < 256 && ok; ++k) {
            uint32_t lo = 0, hi = INF; // q(lo) < k <= q(hi)
            while (hi - lo > 1) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (quantise_gamma(f_of(mid)) >= k)
                    hi = mid;
                else
                    lo = mid;
            }
            tb[k] = hi;
            t_thr[k] = f_of(hi);
            ok = ok && tb[k] >= tb[k - 1];
        }
        auto lookup = [&](float y) {
            int lo = 0, hi = 256; // largest k with thr[k] <= y
            while (hi - lo > 1) {
                const int mid = (lo + hi) / 2;
                if (y >= t_thr[mid])
                    lo = mid;
                else
                    hi = mid;
            }
            return lo;
        };
        for (int k = 1; k < 256 && ok; ++k) // every float within 512 ulps of a threshold
            for (int64_t b = std::max<int64_t>(0, (int64_t)tb[k] - 512); b <= std::min<int64_t>(INF, (int64_t)tb[k] + 512) && ok; ++b)
                ok = lookup(f_of((uint32_t)b)) == quantise_gamma(f_of((uint32_t)b));
        uint64_t st = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < 2000000 && ok; ++i) { // sweep: all exponents, random mantissas
            st ^= st << 13;
            st ^= st >> 7;
            st ^= st << 17;
            const uint32_t b = (uint32_t)(st >> 33) % (INF + 1u);
            ok = lookup(f_of(b)) == quantise_gamma(f_of(b));
        }
        t_special[0] = quantise_gamma(std::nanf(""));
        t_special[1] = quantise_gamma(-1.0f);
        t_special[2] = quantise_gamma(-f_of(INF));
        ok = ok && quantise_gamma(-1.0e-30f) == t_special[1] && quantise_gamma(-3.0e38f) == t_special[1] && quantise_gamma(-0.0f) == 0;
    }
    std::memcpy(thr, t_thr, sizeof(t_thr));
    std::memcpy(special, t_special, sizeof(t_special));
    return ok;
}
} // namespace rt

extern "C" int rt_film_table(float thr[256], uint32_t special[3]) {
    if (!thr || !special)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_film_table: null argument");
    return rt::film_table(thr, special) ? RT_OK : rt::fail(RT_ERR_UNSUPPORTED, "rt_film_table: powf monotonicity check failed");
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

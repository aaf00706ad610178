// hdr_decode.cpp — Radiance RGBE (.hdr) pictures as Texture::load_img sees them (geometry.h:584-598). The reference's default
// environment map is "env.hdr" (config.h:38) and it is loaded with the 8-bit stbi_load, so what reaches the texture is stb_image's
// LOW dynamic range conversion of the picture: per colour channel (uint8)(powf-in-double(v, 1 / 2.2f) * 255 + 0.5f) clamped to
// [0, 255], alpha 255. Restated from the file format (Ward, "Real Pixels", Graphics Gems II) and stb_image v2.30's documented
// behaviour; pinned to the reference's stb build by tests/golden/envmap (ref_probe texture):
//   * header: first line "#?RADIANCE" or "#?RGBE"; lines up to the first empty one, one of which must be FORMAT=32-bit_rle_rgbe;
//     then "-Y <height> +X <width>" (the only orientation stb_image accepts);
//   * pixels: widths 8 .. 32767 may be run-length encoded per scanline (2 2 hi lo, then each of the four components as runs
//     (count > 128: count - 128 copies of the next byte) and dumps (count bytes)); a first scanline that does not start with 2 2
//     switches the WHOLE picture to flat r g b e quadruples (that quirk of stb_image is kept); other widths are flat;
//   * value: mantissa * 2^(e - 136), e = 0 -> black.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rt_host.h"
#include "../rt_error.h"

namespace {

struct HdrError {
    std::string msg;
};

struct Reader {
    const std::vector<uint8_t> &f;
    size_t pos = 0;
    explicit Reader(const std::vector<uint8_t> &file) : f(file) {}
    bool eof() const { return pos >= f.size(); }
    int get8() { return pos < f.size() ? f[pos++] : 0; } // stb_image returns 0 at the end of the data
    // a header line: up to '\n' (not included), at most 1023 characters, the rest of an over-long line is dropped. As in stb_image the
    // end of the data is looked at AFTER a character was taken, so a line that ends exactly with the file loses its last character
    // (only a file without pixel data can notice).
    std::string line() {
        std::string s;
        int c = get8();
        while (!eof() && c != '\n') {
            s.push_back((char)c);
            if (s.size() == 1023) {
                while (!eof() && get8() != '\n') {
                }
                break;
            }
            c = get8();
        }
        return s;
    }
};

uint8_t to_ldr(float v) { // stbi__hdr_to_ldr with the default gamma 2.2 and scale 1
    float z = (float)std::pow((double)(v * 1.0f), (double)(1.0f / 2.2f)) * 255 + 0.5f;
    if (z < 0)
        z = 0;
    if (z > 255)
        z = 255;
    return (uint8_t)(int)z;
}

void put_pixel(uint8_t *out, const uint8_t rgbe[4]) {
    if (rgbe[3] != 0) {
        const float f1 = (float)std::ldexp(1.0f, (int)rgbe[3] - (128 + 8));
        out[0] = to_ldr(rgbe[0] * f1);
        out[1] = to_ldr(rgbe[1] * f1);
        out[2] = to_ldr(rgbe[2] * f1);
    } else {
        out[0] = out[1] = out[2] = to_ldr(0.0f);
    }
    out[3] = 255; // alpha 1.0f * 255 + 0.5f
}

} // namespace

extern "C" int rt_hdr_decode_file(const char *path, uint32_t *w_out, uint32_t *h_out, uint8_t **rgba_out) {
    if (!path || !w_out || !h_out || !rgba_out)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_hdr_decode_file: null argument");
    std::vector<uint8_t> file;
    if (FILE *fp = std::fopen(path, "rb")) {
        uint8_t buf[1 << 16];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, fp)) > 0)
            file.insert(file.end(), buf, buf + n);
        std::fclose(fp);
    } else {
        return rt::fail(RT_ERR_IO, std::string("Failed to load image from ") + path);
    }
    uint8_t *px = nullptr;
    try {
        Reader r(file);
        const std::string first = r.line();
        if (first != "#?RADIANCE" && first != "#?RGBE")
            throw HdrError{"not a Radiance HDR picture"};
        bool valid = false;
        for (;;) {
            const std::string l = r.line();
            if (l.empty())
                break;
            if (l == "FORMAT=32-bit_rle_rgbe")
                valid = true;
        }
        if (!valid)
            throw HdrError{"unsupported HDR format (FORMAT=32-bit_rle_rgbe expected)"};
        const std::string dims = r.line();
        if (dims.compare(0, 3, "-Y ") != 0)
            throw HdrError{"unsupported HDR data layout (-Y <height> +X <width> expected)"};
        const char *p = dims.c_str() + 3;
        char *end = nullptr;
        const long height = std::strtol(p, &end, 10);
        while (*end == ' ')
            ++end;
        if (std::strncmp(end, "+X ", 3) != 0)
            throw HdrError{"unsupported HDR data layout (-Y <height> +X <width> expected)"};
        const long width = std::strtol(end + 3, nullptr, 10);
        if (height <= 0 || width <= 0 || height > (1 << 24) || width > (1 << 24) || (uint64_t)width * (uint64_t)height > (1ull << 28))
            throw HdrError{"HDR picture too large or empty"};
        // The data must be able to hold the picture: flat pixels take 4 bytes each, and the run-length code spends at least 8 bytes per 127
        // pixels (a maximal run per channel), i.e. no valid file carries more than 16 pixels per byte. Without this a header of a few
        // dozen bytes could claim 2^28 pixels: a 1 GiB allocation and, as get8() returns 0 at the end of the data like stb_image's,
        // seconds of pow() over pixels that do not exist.
        if ((uint64_t)width * (uint64_t)height > 16ull * (uint64_t)(file.size() - r.pos) + 16ull)
            throw HdrError{"HDR picture larger than its data can hold"};
        px = static_cast<uint8_t *>(std::malloc((size_t)width * height * 4));
        if (!px)
            return rt::fail(RT_ERR_OOM, "HDR: out of memory");
        auto flat_from = [&](long j0, long i0) { // the rest of the picture as r g b e quadruples
            for (long j = j0; j < height; ++j)
                for (long i = j == j0 ? i0 : 0; i < width; ++i) {
                    uint8_t q[4];
                    for (int k = 0; k < 4; ++k)
                        q[k] = (uint8_t)r.get8();
                    put_pixel(px + ((size_t)j * width + i) * 4, q);
                }
        };
        if (width < 8 || width >= 32768) {
            flat_from(0, 0);
        } else {
            std::vector<uint8_t> scan((size_t)width * 4);
            for (long j = 0; j < height; ++j) {
                const int c1 = r.get8(), c2 = r.get8();
                int len = r.get8();
                if (c1 != 2 || c2 != 2 || (len & 0x80)) {
                    // not run-length encoded: these bytes are the first pixel and everything is flat — from the top of the picture,
                    // whichever scanline said so (stb_image restarts at row 0; for a well-formed flat file this IS row 0)
                    const uint8_t q[4] = {(uint8_t)c1, (uint8_t)c2, (uint8_t)len, (uint8_t)r.get8()};
                    put_pixel(px, q);
                    flat_from(0, 1);
                    break;
                }
                len = (len << 8) | r.get8();
                if (len != width)
                    throw HdrError{"corrupt HDR: invalid decoded scanline length"};
                for (int k = 0; k < 4; ++k) {
                    long i = 0, nleft;
                    while ((nleft = width - i) > 0) {
                        int count = r.get8();
                        if (count > 128) { // run
                            const int value = r.get8();
                            count -= 128;
                            if (count == 0 || count > nleft)
                                throw HdrError{"corrupt HDR: bad run-length data"};
                            for (int z = 0; z < count; ++z)
                                scan[(size_t)(i++) * 4 + k] = (uint8_t)value;
                        } else { // dump
                            if (count == 0 || count > nleft)
                                throw HdrError{"corrupt HDR: bad run-length data"};
                            for (int z = 0; z < count; ++z)
                                scan[(size_t)(i++) * 4 + k] = (uint8_t)r.get8();
                        }
                    }
                }
                for (long i = 0; i < width; ++i)
                    put_pixel(px + ((size_t)j * width + i) * 4, &scan[(size_t)i * 4]);
            }
        }
        *w_out = (uint32_t)width;
        *h_out = (uint32_t)height;
        *rgba_out = px;
    } catch (const HdrError &e) {
        std::free(px);
        return rt::fail(RT_ERR_FORMAT, std::string(path) + ": " + e.msg);
    } catch (const std::bad_alloc &) {
        std::free(px);
        return rt::fail(RT_ERR_OOM, "HDR: out of memory");
    }
    return RT_OK;
}

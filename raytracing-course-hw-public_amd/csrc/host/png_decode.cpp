// png_decode.cpp — minimal PNG -> RGBA8 decoder on top of zlib (host, one-off I/O).
// The reference decodes textures with the vendored stb_image v2.30 forcing 4 channels (geometry.h:584-598).
// Supported here: 8-bit depth, colour types 0/2/3/4/6, non-interlaced, all five filter types, tRNS for
// palettes — which covers what stb would return for such files (grey -> g,g,g,255; RGB -> r,g,b,255).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>

#include "../../../include/rt_host.h"
#include "../rt_error.h"

namespace {

uint32_t be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc)
        return a;
    return pb <= pc ? b : c;
}

} // namespace

extern "C" void rt_free(void *p) { std::free(p); }

extern "C" int rt_png_decode_file(const char *path, uint32_t *w_out, uint32_t *h_out, uint8_t **rgba_out) {
    if (!path || !w_out || !h_out || !rgba_out)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_png_decode_file: null argument");
    FILE *f = std::fopen(path, "rb");
    if (!f)
        return rt::fail(RT_ERR_IO, std::string("Failed to load image from ") + path); // geometry.h:588
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0)
        file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0)
        return rt::fail(RT_ERR_FORMAT, std::string("not a PNG file: ") + path);
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t p = 8;
    bool seen_ihdr = false;
    while (p + 12 <= file.size()) {
        uint32_t len = be32(&file[p]);
        const uint8_t *type = &file[p + 4];
        if (p + 12 + (size_t)len > file.size())
            return rt::fail(RT_ERR_FORMAT, "PNG: truncated chunk");
        const uint8_t *data = &file[p + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13)
                return rt::fail(RT_ERR_FORMAT, "PNG: bad IHDR");
            w = be32(data);
            h = be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
            seen_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        p += 12 + (size_t)len;
    }
    if (!seen_ihdr || w == 0 || h == 0)
        return rt::fail(RT_ERR_FORMAT, "PNG: missing IHDR");
    // stb_image caps both dimensions at 1 << 24 (STBI_MAX_DIMENSIONS) and rejects images whose byte count overflows; the
    // sizes below are computed in 64 bits and bounded before anything is allocated or indexed
    if (w > (1u << 24) || h > (1u << 24) || (uint64_t)w * h > (1ull << 28))
        return rt::fail(RT_ERR_FORMAT, "PNG: image too large (limit 2^24 per side, 2^28 pixels)");
    if (depth != 8 || interlace != 0)
        return rt::fail(RT_ERR_FORMAT, "PNG: only 8-bit non-interlaced images are supported");
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch)
        return rt::fail(RT_ERR_FORMAT, "PNG: unknown colour type");
    size_t stride = (size_t)w * ch;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = raw.size();
    int zr = uncompress(raw.data(), &raw_len, idat.data(), idat.size());
    if (zr != Z_OK || raw_len != raw.size())
        return rt::fail(RT_ERR_FORMAT, "PNG: inflate failed");
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *src = &raw[(stride + 1) * y];
        uint8_t ft = src[0];
        ++src;
        uint8_t *dst = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= (size_t)ch ? dst[i - ch] : 0;
            int b = up ? up[i] : 0;
            int c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int v = src[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return rt::fail(RT_ERR_FORMAT, "PNG: bad filter type");
            }
            dst[i] = (uint8_t)v;
        }
    }
    uint8_t *out = (uint8_t *)std::malloc((size_t)w * h * 4);
    if (!out)
        return rt::fail(RT_ERR_OOM, "PNG: out of memory");
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t *s = &img[i * ch];
        uint8_t *d = &out[i * 4];
        switch (ctype) {
        case 0: d[0] = d[1] = d[2] = s[0]; d[3] = 255; break;
        case 2: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 255; break;
        case 3: {
            size_t k = s[0];
            if (3 * k + 2 >= plte.size()) {
                d[0] = d[1] = d[2] = 0;
            } else {
                d[0] = plte[3 * k];
                d[1] = plte[3 * k + 1];
                d[2] = plte[3 * k + 2];
            }
            d[3] = k < trns.size() ? trns[k] : 255;
            break;
        }
        case 4: d[0] = d[1] = d[2] = s[0]; d[3] = s[1]; break;
        default: d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3]; break;
        }
    }
    *w_out = w;
    *h_out = h;
    *rgba_out = out;
    return RT_OK;
}

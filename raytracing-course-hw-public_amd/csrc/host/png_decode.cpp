// png_decode.cpp — PNG -> RGBA8 decoder on top of zlib (host, one-off I/O).
//
// The reference decodes textures with the vendored stb_image v2.30, forcing 4 channels (geometry.h:584-598:
// stbi_load(path, &w, &h, &n, 4), then / 255.0f). This decoder covers the whole PNG format the way stb_image answers it,
// pinned by fixtures whose expected texels come from the reference's own stb build (oracle/ref_probe "texture" mode ->
// tests/golden/png/, tests/test_png_golden.py):
//   * colour types 0 (grey), 2 (RGB), 3 (palette), 4 (grey + alpha), 6 (RGBA); bit depths 1, 2, 4, 8, 16; Adam7 interlace;
//   * grey samples below 8 bits are scaled to 0..255 (x255, x85, x17); 16-bit samples keep their HIGH byte;
//   * tRNS: per-index alpha for palettes; for grey / RGB a colour key (compared on the 16-bit samples of a 16-bit image,
//     on the scaled low byte otherwise) whose pixels get alpha 0;
//   * 4 output channels always: grey -> g,g,g,a and RGB -> r,g,b,a with a = 255 unless the file says otherwise;
//   * chunk CRCs are not verified (stb_image skips them); an unknown CRITICAL chunk is an error, ancillary ones are skipped;
//   * dimensions above 2^24 (STBI_MAX_DIMENSIONS) or 2^28 pixels are refused before anything is allocated.
// JPEG is read by jpeg_decode.cpp (rt_image_decode_file picks the decoder by signature). BMP / TGA / GIF / PSD / HDR / PNM,
// which stb_image would also read, are NOT supported: the error message names the format so that a user knows to convert
// the texture (INTEGRATION.md "Loader differences").
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
uint32_t be16(const uint8_t *p) { return (uint32_t(p[0]) << 8) | p[1]; }

int paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc)
        return a;
    return pb <= pc ? b : c;
}

const char *sniff_other_format(const std::vector<uint8_t> &f) {
    if (f.size() >= 3 && f[0] == 0xFF && f[1] == 0xD8 && f[2] == 0xFF)
        return "JPEG";
    if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M')
        return "BMP";
    if (f.size() >= 6 && (!std::memcmp(f.data(), "GIF87a", 6) || !std::memcmp(f.data(), "GIF89a", 6)))
        return "GIF";
    if (f.size() >= 4 && !std::memcmp(f.data(), "8BPS", 4))
        return "PSD";
    if (f.size() >= 10 && (!std::memcmp(f.data(), "#?RADIANCE", 10) || !std::memcmp(f.data(), "#?RGBE", 6)))
        return "Radiance HDR";
    if (f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6'))
        return "PNM";
    return nullptr;
}

// One (sub)image: `raw` holds h scanlines of [filter byte][row bytes]; writes w*h*ch samples of `bytes` bytes each
// (big-endian for 16 bit) into `out`, samples below 8 bits unpacked to one byte each (unscaled).
bool unfilter_image(const uint8_t *raw, size_t raw_len, size_t &used, uint32_t w, uint32_t h, int ch, int depth, std::vector<uint8_t> &out, std::string &err) {
    const size_t row_bytes = ((size_t)w * ch * depth + 7) / 8;
    const int bpp = depth < 8 ? 1 : ch * depth / 8; // filter distance in bytes
    if (raw_len < (row_bytes + 1) * h) {
        err = "PNG: not enough pixel data";
        return false;
    }
    std::vector<uint8_t> prev(row_bytes, 0), cur(row_bytes);
    const int sample_bytes = depth == 16 ? 2 : 1;
    out.assign((size_t)w * h * ch * sample_bytes, 0);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *src = raw + (row_bytes + 1) * y;
        const uint8_t ft = src[0];
        ++src;
        if (ft > 4) {
            err = "PNG: bad filter type";
            return false;
        }
        for (size_t i = 0; i < row_bytes; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0;
            const int b = prev[i];
            const int c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int v = src[i];
            switch (ft) {
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: break;
            }
            cur[i] = (uint8_t)v;
        }
        uint8_t *dst = &out[(size_t)y * w * ch * sample_bytes];
        if (depth >= 8) {
            std::memcpy(dst, cur.data(), row_bytes);
        } else { // unpack 1 / 2 / 4-bit samples, most significant bits first
            const int per_byte = 8 / depth, mask = (1 << depth) - 1;
            for (size_t s = 0; s < (size_t)w * ch; ++s)
                dst[s] = (uint8_t)((cur[s / per_byte] >> (8 - depth - (int)(s % per_byte) * depth)) & mask);
        }
        prev.swap(cur);
    }
    used = (row_bytes + 1) * h;
    return true;
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
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) {
        if (const char *other = sniff_other_format(file))
            return rt::fail(RT_ERR_FORMAT, std::string(path) + ": " + other + " image: this loader reads PNG and JPEG textures only (stb_image, which the reference uses, would read it; convert the texture)");
        return rt::fail(RT_ERR_FORMAT, std::string("not a PNG file: ") + path);
    }
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t p = 8;
    bool seen_ihdr = false, seen_iend = false;
    while (p + 12 <= file.size()) {
        uint32_t len = be32(&file[p]);
        const uint8_t *type = &file[p + 4];
        if (p + 12 + (size_t)len > file.size())
            return rt::fail(RT_ERR_FORMAT, "PNG: truncated chunk");
        const uint8_t *data = &file[p + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13 || seen_ihdr)
                return rt::fail(RT_ERR_FORMAT, "PNG: bad IHDR");
            w = be32(data);
            h = be32(data + 4);
            depth = data[8];
            ctype = data[9];
            if (data[10] != 0 || data[11] != 0)
                return rt::fail(RT_ERR_FORMAT, "PNG: bad compression / filter method");
            interlace = data[12];
            seen_ihdr = true;
        } else if (!seen_ihdr) {
            return rt::fail(RT_ERR_FORMAT, "PNG: first chunk is not IHDR");
        } else if (!std::memcmp(type, "PLTE", 4)) {
            if (len > 256 * 3 || len % 3 != 0)
                return rt::fail(RT_ERR_FORMAT, "PNG: bad PLTE");
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            seen_iend = true;
            break;
        } else if ((type[0] & 0x20) == 0) { // upper-case first letter: a critical chunk this decoder does not know
            return rt::fail(RT_ERR_FORMAT, std::string("PNG: unknown critical chunk ") + std::string(reinterpret_cast<const char *>(type), 4));
        }
        p += 12 + (size_t)len;
    }
    (void)seen_iend;
    if (!seen_ihdr || w == 0 || h == 0)
        return rt::fail(RT_ERR_FORMAT, "PNG: missing IHDR");
    // stb_image caps both dimensions at 1 << 24 (STBI_MAX_DIMENSIONS) and rejects images whose byte count overflows; the
    // sizes below are computed in 64 bits and bounded before anything is allocated or indexed
    if (w > (1u << 24) || h > (1u << 24) || (uint64_t)w * h > (1ull << 28))
        return rt::fail(RT_ERR_FORMAT, "PNG: image too large (limit 2^24 per side, 2^28 pixels)");
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch)
        return rt::fail(RT_ERR_FORMAT, "PNG: unknown colour type");
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                                     : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!depth_ok)
        return rt::fail(RT_ERR_FORMAT, "PNG: bit depth " + std::to_string(depth) + " is not valid for colour type " + std::to_string(ctype));
    if (interlace > 1)
        return rt::fail(RT_ERR_FORMAT, "PNG: unknown interlace method");
    if (ctype == 3 && plte.empty())
        return rt::fail(RT_ERR_FORMAT, "PNG: palette image without PLTE");
    if (!trns.empty()) {
        if (ctype == 4 || ctype == 6)
            return rt::fail(RT_ERR_FORMAT, "PNG: tRNS with an alpha channel");
        if (ctype == 3 ? trns.size() > plte.size() / 3 : trns.size() != (size_t)ch * 2)
            return rt::fail(RT_ERR_FORMAT, "PNG: bad tRNS length");
    }

    // ---- inflate
    const uint64_t bits_pp = (uint64_t)ch * depth;
    uint64_t raw_need = 0;
    if (!interlace) {
        raw_need = (((uint64_t)w * bits_pp + 7) / 8 + 1) * h;
    } else {
        static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int k = 0; k < 7; ++k) {
            const uint64_t pw = (w - xs[k] + dx[k] - 1) / dx[k], ph = (h - ys[k] + dy[k] - 1) / dy[k];
            if (w > (uint32_t)xs[k] && h > (uint32_t)ys[k] && pw && ph)
                raw_need += ((pw * bits_pp + 7) / 8 + 1) * ph;
        }
    }
    std::vector<uint8_t> raw(raw_need);
    uLongf raw_len = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if ((zr != Z_OK && zr != Z_BUF_ERROR) || raw_len < raw.size())
        return rt::fail(RT_ERR_FORMAT, "PNG: inflate failed");

    // ---- unfilter (+ de-interlace) into samples: w*h*ch entries of 1 byte (depth <= 8, unscaled) or 2 bytes (depth 16)
    const int sb = depth == 16 ? 2 : 1;
    std::vector<uint8_t> samples((size_t)w * h * ch * sb, 0);
    std::string err;
    if (!interlace) {
        size_t used = 0;
        if (!unfilter_image(raw.data(), raw.size(), used, w, h, ch, depth, samples, err))
            return rt::fail(RT_ERR_FORMAT, err);
    } else {
        static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
        size_t off = 0;
        std::vector<uint8_t> pass;
        for (int k = 0; k < 7; ++k) {
            if (w <= (uint32_t)xs[k] || h <= (uint32_t)ys[k])
                continue;
            const uint32_t pw = (w - xs[k] + dx[k] - 1) / dx[k], ph = (h - ys[k] + dy[k] - 1) / dy[k];
            if (!pw || !ph)
                continue;
            size_t used = 0;
            if (!unfilter_image(raw.data() + off, raw.size() - off, used, pw, ph, ch, depth, pass, err))
                return rt::fail(RT_ERR_FORMAT, err);
            off += used;
            const size_t px = (size_t)ch * sb;
            for (uint32_t y = 0; y < ph; ++y)
                for (uint32_t x = 0; x < pw; ++x)
                    std::memcpy(&samples[(((size_t)(ys[k] + y * dy[k])) * w + (xs[k] + x * dx[k])) * px], &pass[((size_t)y * pw + x) * px], px);
        }
    }

    // ---- to RGBA8 the way stb_image does with req_comp = 4
    uint8_t *out = (uint8_t *)std::malloc((size_t)w * h * 4);
    if (!out)
        return rt::fail(RT_ERR_OOM, "PNG: out of memory");
    static const int depth_scale[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
    const int scale = ctype == 0 && depth < 8 ? depth_scale[depth] : 1;
    uint32_t key16[3] = {0, 0, 0};
    uint8_t key8[3] = {0, 0, 0};
    const bool has_key = !trns.empty() && ctype != 3;
    if (has_key)
        for (int k = 0; k < ch; ++k) {
            key16[k] = be16(&trns[2 * k]);
            key8[k] = (uint8_t)((key16[k] & 255u) * (uint32_t)(depth < 8 ? depth_scale[depth] : 1));
        }
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        uint8_t *d = &out[i * 4];
        if (ctype == 3) {
            const size_t k = samples[i];
            if (3 * k + 2 >= plte.size()) {
                d[0] = d[1] = d[2] = 0;
            } else {
                d[0] = plte[3 * k];
                d[1] = plte[3 * k + 1];
                d[2] = plte[3 * k + 2];
            }
            d[3] = k < trns.size() ? trns[k] : 255;
            continue;
        }
        uint8_t v[4] = {0, 0, 0, 255};
        bool keyed = has_key;
        for (int k = 0; k < ch; ++k) {
            if (depth == 16) {
                const uint32_t s16 = be16(&samples[(i * ch + k) * 2]);
                v[k] = (uint8_t)(s16 >> 8); // stb_image keeps the high byte of a 16-bit sample
                if (has_key && k < (ctype == 0 ? 1 : 3) && s16 != key16[k])
                    keyed = false;
            } else {
                v[k] = (uint8_t)(samples[i * ch + k] * scale);
                if (has_key && k < (ctype == 0 ? 1 : 3) && v[k] != key8[k])
                    keyed = false;
            }
        }
        switch (ctype) {
        case 0: d[0] = d[1] = d[2] = v[0]; d[3] = keyed ? 0 : 255; break;
        case 2: d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = keyed ? 0 : 255; break;
        case 4: d[0] = d[1] = d[2] = v[0]; d[3] = v[1]; break;
        default: d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3]; break;
        }
    }
    *w_out = w;
    *h_out = h;
    *rgba_out = out;
    return RT_OK;
}

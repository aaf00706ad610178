// jpeg_decode.cpp — JPEG (baseline, extended-sequential and progressive Huffman, 8-bit) -> RGBA8 (host, one-off I/O).
//
// The reference decodes textures with the vendored stb_image v2.30, forcing 4 channels (geometry.h:584-598). Real glTF
// assets carry .jpg textures (Khronos Sponza does), so the host loader reads them too — and returns the SAME BYTES stb_image
// returns, because every texel feeds the bit-exact parity chain. The entropy decoding follows ITU-T T.81 (Annex F baseline,
// Annex G progressive: spectral selection, successive approximation, EOB runs, restart intervals) and has one possible
// answer; the parts of a JPEG decoder that are NOT standardised are restated here the way stb_image does them, pinned by
// fixtures decoded through the reference's own stb build (oracle/_ref/ref_probe "texture"; tests/golden/jpeg/,
// tests/test_jpeg_golden.py):
//   * inverse DCT: the Loeffler-Ligtenberg-Moschytz 1-D transform (IJG "islow") in 32-bit integers with 12-bit constants
//     round(c * 4096), two extra bits kept after the column pass (+512 >> 10), +65536 + (128 << 17) >> 17 after the row pass,
//     clamped to 0..255;
//   * chroma upsampling: 2x1 / 1x2 by (3 * near + far + 2) >> 2 with replicated edges, 2x2 by the separable form of the same
//     filter ((3 * t0 + t1 + 8) >> 4 on vertical sums t = 3 * near + far), any other ratio by replication; the vertical
//     neighbour of an output row is the next stored row for the lower half of a sample and the previous one for the upper half;
//   * YCbCr -> RGB in 20-bit fixed point: y' = (y << 20) + (1 << 19); r = y' + cr * K(1.40200); g = y' - cr * K(0.71414) +
//     ((-cb * K(0.34414)) & 0xffff0000); b = y' + cb * K(1.77200), K(x) = round(x * 4096) << 8; >> 20, clamped;
//   * three components are YCbCr unless their ids are 'R','G','B' or an Adobe APP14 marker says "no transform" in a file
//     without a JFIF header; one component is grey (g, g, g); alpha is 255. Four components (CMYK / YCCK) are refused.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rt_host.h"
#include "../rt_error.h"

namespace {

struct JpegError {
    std::string msg;
};
[[noreturn]] void bad(const std::string &m) { throw JpegError{"JPEG: " + m}; }

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman { // T.81 F.2.2.3 decoding tables
    bool present = false;
    uint8_t vals[256];
    int mincode[17], maxcode[18], valptr[17];
    void build(const uint8_t counts[16], const uint8_t *symbols, int n) {
        std::memcpy(vals, symbols, (size_t)n);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len))
                bad("bad Huffman code lengths");
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        present = true;
    }
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    int marker = 0; // a marker met inside the entropy-coded data (0 = none): from then on zero bits are supplied
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    void fill() {
        while (nbits <= 24) {
            uint32_t byte = 0;
            if (!marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    uint32_t nxt = p < end ? *p : 0xD9;
                    while (nxt == 0xFF && p + 1 < end) // fill bytes
                        nxt = *++p;
                    if (nxt == 0) {
                        ++p; // stuffed zero
                    } else {
                        marker = (int)nxt;
                        ++p;
                        byte = 0;
                    }
                }
            }
            acc |= byte << (24 - nbits);
            nbits += 8;
        }
    }
    int bit() {
        if (nbits < 1)
            fill();
        const int b = (int)(acc >> 31);
        acc <<= 1;
        --nbits;
        return b;
    }
    int bits(int n) {
        if (n == 0)
            return 0;
        if (nbits < n)
            fill();
        const int v = (int)(acc >> (32 - n));
        acc <<= n;
        nbits -= n;
        return v;
    }
    int decode(const Huffman &h) {
        if (!h.present)
            bad("scan uses an undefined Huffman table");
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (code << 1) | bit();
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len])
                return h.vals[h.valptr[len] + code - h.mincode[len]];
        }
        bad("bad Huffman code");
    }
    void reset() {
        acc = 0;
        nbits = 0;
        marker = 0;
    }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; } // T.81 F.2.2.1

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;     // Huffman table selectors of the current scan
    int x = 0, y = 0;       // size in samples
    int w2 = 0, h2 = 0;     // allocated plane size (whole MCUs)
    int bw = 0, bh = 0;     // plane size in blocks
    int dc_pred = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coef; // progressive: bw * bh * 64, natural order
};

// round(c * 4096) as (int)(c * 4096 + 0.5): for the NEGATIVE constants this truncates toward zero, i.e. -7567 for -1.847759065,
// not -7568 — the constants below are written with their signs for that reason (they are what stb_image's decoder uses)
constexpr int fx(double x) { return (int)(x * 4096 + 0.5); }

// The IDCT's 32-bit integers with WRAPPING arithmetic: valid pictures never come near the range, but the coefficients of a corrupt file can
// overflow the sums and products, which is undefined for plain int (found by tools/fuzz_loaders.py under UBSan). Two's-complement wrap-around
// is what the reference's stb_image build does on this hardware, and it is defined behaviour here.
struct wi {
    int32_t v;
    wi() : v(0) {}
    wi(int x) : v(x) {}
    friend wi operator+(wi a, wi b) { return wi((int32_t)((uint32_t)a.v + (uint32_t)b.v)); }
    friend wi operator-(wi a, wi b) { return wi((int32_t)((uint32_t)a.v - (uint32_t)b.v)); }
    friend wi operator*(wi a, wi b) { return wi((int32_t)((uint32_t)a.v * (uint32_t)b.v)); }
    wi &operator+=(wi b) { return *this = *this + b; }
    friend int operator>>(wi a, int k) { return a.v >> k; } // arithmetic shift (C++20)
};

// 1-D LL&M inverse DCT on (s0..s7), results in x0..x3 / t0..t3 as the even / odd halves
#define RT_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                   \
    wi t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                           \
    p2 = s2;                                                                         \
    p3 = s6;                                                                         \
    p1 = (p2 + p3) * fx(0.5411961f);                                                 \
    t2 = p1 + p3 * fx(-1.847759065f);                                                 \
    t3 = p1 + p2 * fx(0.765366865f);                                                 \
    p2 = s0;                                                                         \
    p3 = s4;                                                                         \
    t0 = (p2 + p3) * 4096;                                                           \
    t1 = (p2 - p3) * 4096;                                                           \
    x0 = t0 + t3;                                                                    \
    x3 = t0 - t3;                                                                    \
    x1 = t1 + t2;                                                                    \
    x2 = t1 - t2;                                                                    \
    t0 = s7;                                                                         \
    t1 = s5;                                                                         \
    t2 = s3;                                                                         \
    t3 = s1;                                                                         \
    p3 = t0 + t2;                                                                    \
    p4 = t1 + t3;                                                                    \
    p1 = t0 + t3;                                                                    \
    p2 = t1 + t2;                                                                    \
    p5 = (p3 + p4) * fx(1.175875602f);                                               \
    t0 = t0 * fx(0.298631336f);                                                      \
    t1 = t1 * fx(2.053119869f);                                                      \
    t2 = t2 * fx(3.072711026f);                                                      \
    t3 = t3 * fx(1.501321110f);                                                      \
    p1 = p5 + p1 * fx(-0.899976223f);                                                 \
    p2 = p5 + p2 * fx(-2.562915447f);                                                 \
    p3 = p3 * fx(-1.961570560f);                                                      \
    p4 = p4 * fx(-0.390180644f);                                                      \
    t3 += p1 + p4;                                                                   \
    t2 += p2 + p3;                                                                   \
    t1 += p2 + p4;                                                                   \
    t0 += p1 + p3;

inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

void idct_block(uint8_t *out, int stride, const int16_t d[64]) {
    int val[64];
    for (int i = 0; i < 8; ++i) {
        RT_IDCT_1D(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i])
        x0 += wi(512), x1 += wi(512), x2 += wi(512), x3 += wi(512);
        val[i] = (x0 + t3) >> 10;
        val[56 + i] = (x0 - t3) >> 10;
        val[8 + i] = (x1 + t2) >> 10;
        val[48 + i] = (x1 - t2) >> 10;
        val[16 + i] = (x2 + t1) >> 10;
        val[40 + i] = (x2 - t1) >> 10;
        val[24 + i] = (x3 + t0) >> 10;
        val[32 + i] = (x3 - t0) >> 10;
    }
    for (int i = 0; i < 8; ++i) {
        const int *v = val + 8 * i;
        uint8_t *o = out + (size_t)stride * i;
        RT_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        const int bias = 65536 + (128 << 17);
        x0 += wi(bias), x1 += wi(bias), x2 += wi(bias), x3 += wi(bias);
        o[0] = clamp8((x0 + t3) >> 17);
        o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17);
        o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17);
        o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17);
        o[4] = clamp8((x3 - t0) >> 17);
    }
}

struct Decoder {
    const uint8_t *data, *end;
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false, jfif = false, seen_sof = false;
    int adobe_transform = -1;
    int hmax = 1, vmax = 1, mcu_w = 8, mcu_h = 8, mcus_x = 0, mcus_y = 0;
    Component comp[4];
    uint16_t quant[4][64]; // natural order
    bool quant_present[4] = {false, false, false, false};
    Huffman dc[4], ac[4];
    int restart_interval = 0;
    // current scan
    int scan_n = 0, order[4] = {0, 0, 0, 0};
    int ss = 0, se = 63, ah = 0, al = 0;
    int eob_run = 0;

    static int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

    void frame_header(const uint8_t *p, int len) {
        if (seen_sof)
            bad("more than one frame header");
        if (len < 6 || p[0] != 8)
            bad("only 8-bit samples are supported");
        height = be16(p + 1);
        width = be16(p + 3);
        ncomp = p[5];
        if (height == 0)
            bad("image height 0 (DNL) is not supported");
        if (width == 0)
            bad("image width 0");
        if (width > (1 << 24) || height > (1 << 24) || (uint64_t)width * height > (1ull << 28))
            bad("image too large (limit 2^24 per side, 2^28 pixels)");
        if (ncomp == 4)
            bad("4-component (CMYK / YCCK) images are not supported");
        if (ncomp != 1 && ncomp != 3)
            bad("bad component count");
        if (len != 6 + 3 * ncomp)
            bad("bad SOF length");
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            c.id = p[6 + 3 * i];
            c.h = p[7 + 3 * i] >> 4;
            c.v = p[7 + 3 * i] & 15;
            c.tq = p[8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
                bad("bad sampling factors / quantisation table id");
            hmax = c.h > hmax ? c.h : hmax;
            vmax = c.v > vmax ? c.v : vmax;
        }
        for (int i = 0; i < ncomp; ++i)
            if (hmax % comp[i].h != 0 || vmax % comp[i].v != 0)
                bad("sampling factors that do not divide the maximum are not supported");
        mcu_w = hmax * 8;
        mcu_h = vmax * 8;
        mcus_x = (width + mcu_w - 1) / mcu_w;
        mcus_y = (height + mcu_h - 1) / mcu_h;
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            c.x = (width * c.h + hmax - 1) / hmax;
            c.y = (height * c.v + vmax - 1) / vmax;
            c.w2 = mcus_x * c.h * 8;
            c.h2 = mcus_y * c.v * 8;
            c.bw = c.w2 / 8;
            c.bh = c.h2 / 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive)
                c.coef.assign((size_t)c.bw * c.bh * 64, 0);
        }
        seen_sof = true;
    }

    void scan_header(const uint8_t *p, int len) {
        if (!seen_sof)
            bad("scan before the frame header");
        if (len < 6) // Ns + one component + Ss, Se, Ah/Al: anything shorter would read past the segment (and the file)
            bad("bad SOS");
        scan_n = p[0];
        if (scan_n < 1 || scan_n > ncomp || len != 4 + 2 * scan_n)
            bad("bad SOS");
        for (int i = 0; i < scan_n; ++i) {
            int which = -1;
            for (int k = 0; k < ncomp; ++k)
                if (comp[k].id == p[1 + 2 * i])
                    which = k;
            if (which < 0)
                bad("scan names an unknown component");
            order[i] = which;
            comp[which].td = p[2 + 2 * i] >> 4;
            comp[which].ta = p[2 + 2 * i] & 15;
            if (comp[which].td > 3 || comp[which].ta > 3)
                bad("bad Huffman table selector");
        }
        ss = p[1 + 2 * scan_n];
        se = p[2 + 2 * scan_n];
        ah = p[3 + 2 * scan_n] >> 4;
        al = p[3 + 2 * scan_n] & 15;
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13 || (ss == 0 && se != 0) || (ss != 0 && scan_n != 1))
                bad("bad progressive scan parameters");
        } else {
            if (ss != 0 || ah != 0 || al != 0)
                bad("bad sequential scan parameters");
            se = 63;
        }
    }

    // ---- one block, sequential (T.81 F.2.2)
    void block_sequential(BitReader &br, Component &c, int16_t out[64]) {
        std::memset(out, 0, 64 * sizeof(int16_t));
        const uint16_t *q = quant[c.tq];
        const int t = br.decode(dc[c.td]);
        if (t > 15)
            bad("bad DC category");
        const int diff = t ? extend(br.bits(t), t) : 0;
        c.dc_pred += diff;
        out[0] = (int16_t)(c.dc_pred * q[0]);
        for (int k = 1; k < 64;) {
            const int rs = br.decode(ac[c.ta]);
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xF0)
                    break; // end of block
                k += 16;
            } else {
                k += r;
                if (k > 63)
                    bad("AC run past the end of the block");
                const int z = ZIGZAG[k++];
                out[z] = (int16_t)(extend(br.bits(s), s) * q[z]);
            }
        }
    }
    // ---- progressive (T.81 G.1.2): DC first / refinement, AC first / refinement on one block of coefficients
    void block_prog_dc(BitReader &br, Component &c, int16_t *d) {
        if (ah == 0) {
            const int t = br.decode(dc[c.td]);
            if (t > 15)
                bad("bad DC category");
            const int diff = t ? extend(br.bits(t), t) : 0;
            c.dc_pred += diff;
            d[0] = (int16_t)(c.dc_pred * (1 << al));
        } else if (br.bit()) {
            d[0] = (int16_t)(d[0] + (1 << al));
        }
    }
    void refine_nonzero(BitReader &br, int16_t *p, int bit) {
        if (br.bit() && (*p & bit) == 0)
            *p = (int16_t)(*p > 0 ? *p + bit : *p - bit);
    }
    void block_prog_ac(BitReader &br, Component &c, int16_t *d) {
        const Huffman &h = ac[c.ta];
        if (ah == 0) {
            if (eob_run) {
                --eob_run;
                return;
            }
            for (int k = ss; k <= se;) {
                const int rs = br.decode(h);
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = (1 << r) - 1;
                        if (r)
                            eob_run += br.bits(r);
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63)
                        bad("AC run past the end of the block");
                    d[ZIGZAG[k++]] = (int16_t)(extend(br.bits(s), s) * (1 << al));
                }
            }
        } else {
            const int bit = 1 << al;
            if (eob_run) {
                --eob_run;
                for (int k = ss; k <= se; ++k) {
                    int16_t *p = &d[ZIGZAG[k]];
                    if (*p != 0)
                        refine_nonzero(br, p, bit);
                }
                return;
            }
            int k = ss;
            do {
                const int rs = br.decode(h);
                int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = (1 << r) - 1;
                        if (r)
                            eob_run += br.bits(r);
                        r = 64; // to the end of the band: only refinements of nonzero coefficients follow
                    }
                } else {
                    if (s != 1)
                        bad("bad refinement symbol");
                    s = br.bit() ? bit : -bit;
                }
                while (k <= se) {
                    int16_t *p = &d[ZIGZAG[k++]];
                    if (*p != 0) {
                        refine_nonzero(br, p, bit);
                    } else {
                        if (r == 0) {
                            *p = (int16_t)s;
                            break;
                        }
                        --r;
                    }
                }
            } while (k <= se);
        }
    }

    void restart(BitReader &br) {
        // byte-align, then the RSTn marker must be the next thing in the stream
        br.nbits = 0;
        br.acc = 0;
        if (!br.marker) {
            br.fill(); // runs into the marker
        }
        if (br.marker < 0xD0 || br.marker > 0xD7)
            bad("restart marker missing");
        br.reset();
        for (int i = 0; i < ncomp; ++i)
            comp[i].dc_pred = 0;
        eob_run = 0;
    }

    const uint8_t *decode_scan(const uint8_t *p) {
        BitReader br(p, end);
        for (int i = 0; i < ncomp; ++i)
            comp[i].dc_pred = 0;
        eob_run = 0;
        int todo = restart_interval ? restart_interval : 0x7FFFFFFF;
        int16_t blk[64];
        auto one_block = [&](Component &c, int bx, int by) {
            if (!progressive) {
                block_sequential(br, c, blk);
                idct_block(&c.plane[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, blk);
            } else {
                int16_t *d = &c.coef[((size_t)by * c.bw + bx) * 64];
                if (ss == 0)
                    block_prog_dc(br, c, d);
                else
                    block_prog_ac(br, c, d);
            }
        };
        if (scan_n == 1) { // non-interleaved: the component's own blocks, row by row
            Component &c = comp[order[0]];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int by = 0; by < h; ++by)
                for (int bx = 0; bx < w; ++bx) {
                    one_block(c, bx, by);
                    if (--todo <= 0) {
                        if (by == h - 1 && bx == w - 1)
                            break;
                        restart(br);
                        todo = restart_interval;
                    }
                }
        } else {
            for (int my = 0; my < mcus_y; ++my)
                for (int mx = 0; mx < mcus_x; ++mx) {
                    for (int i = 0; i < scan_n; ++i) {
                        Component &c = comp[order[i]];
                        for (int v = 0; v < c.v; ++v)
                            for (int h = 0; h < c.h; ++h)
                                one_block(c, mx * c.h + h, my * c.v + v);
                    }
                    if (--todo <= 0) {
                        if (my == mcus_y - 1 && mx == mcus_x - 1)
                            break;
                        restart(br);
                        todo = restart_interval;
                    }
                }
        }
        // where the entropy-coded segment ends: at the marker the reader ran into, or search for it
        if (br.marker)
            return br.p - 2;
        const uint8_t *q = br.p;
        while (q + 1 < end) {
            if (q[0] == 0xFF && q[1] != 0 && !(q[1] >= 0xD0 && q[1] <= 0xD7) && q[1] != 0xFF)
                return q;
            ++q;
        }
        return end;
    }

    void finish_progressive() {
        int16_t blk[64];
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            if (!quant_present[c.tq])
                bad("missing quantisation table");
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int by = 0; by < h; ++by)
                for (int bx = 0; bx < w; ++bx) {
                    const int16_t *d = &c.coef[((size_t)by * c.bw + bx) * 64];
                    for (int k = 0; k < 64; ++k)
                        blk[k] = (int16_t)(d[k] * quant[c.tq][k]);
                    idct_block(&c.plane[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, blk);
                }
        }
    }

    void parse() {
        if (end - data < 4 || data[0] != 0xFF || data[1] != 0xD8)
            bad("no SOI marker");
        const uint8_t *p = data + 2;
        bool done = false;
        while (!done) {
            while (p < end && *p != 0xFF)
                ++p; // garbage between segments is skipped, as stb_image does
            while (p < end && *p == 0xFF)
                ++p;
            if (p >= end)
                break;
            const int m = *p++;
            if (m == 0xD9) {
                done = true;
                break;
            }
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7))
                continue; // stand-alone markers
            if (end - p < 2)
                bad("truncated segment");
            const int len = be16(p) - 2;
            if (len < 0 || end - p - 2 < len)
                bad("truncated segment");
            const uint8_t *s = p + 2;
            p += 2 + len;
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                progressive = m == 0xC2;
                frame_header(s, len);
            } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
                bad(m >= 0xC9 ? "arithmetic-coded JPEG is not supported" : "lossless / hierarchical JPEG is not supported");
            } else if (m == 0xC4) { // DHT
                const uint8_t *q = s, *qe = s + len;
                while (q < qe) {
                    if (qe - q < 17)
                        bad("bad DHT");
                    const int tc = q[0] >> 4, th = q[0] & 15;
                    int n = 0;
                    for (int i = 0; i < 16; ++i)
                        n += q[1 + i];
                    if (tc > 1 || th > 3 || n > 256 || qe - q < 17 + n)
                        bad("bad DHT");
                    (tc ? ac : dc)[th].build(q + 1, q + 17, n);
                    q += 17 + n;
                }
            } else if (m == 0xDB) { // DQT, zig-zag order in the file
                const uint8_t *q = s, *qe = s + len;
                while (q < qe) {
                    const int pq = q[0] >> 4, tq = q[0] & 15;
                    if (pq > 1 || tq > 3 || qe - q < 1 + (pq ? 128 : 64))
                        bad("bad DQT");
                    for (int i = 0; i < 64; ++i)
                        quant[tq][ZIGZAG[i]] = (uint16_t)(pq ? be16(q + 1 + 2 * i) : q[1 + i]);
                    quant_present[tq] = true;
                    q += 1 + (pq ? 128 : 64);
                }
            } else if (m == 0xDD) { // DRI
                if (len != 2)
                    bad("bad DRI");
                restart_interval = be16(s);
            } else if (m == 0xE0) {
                if (len >= 5 && !std::memcmp(s, "JFIF\0", 5))
                    jfif = true;
            } else if (m == 0xEE) {
                if (len >= 12 && !std::memcmp(s, "Adobe\0", 6))
                    adobe_transform = s[11];
            } else if (m == 0xDA) { // SOS
                scan_header(s, len);
                if (!progressive)
                    for (int i = 0; i < scan_n; ++i)
                        if (!quant_present[comp[order[i]].tq])
                            bad("missing quantisation table");
                p = decode_scan(p);
            }
            // every other segment (APPn, COM, ...) is skipped
        }
        if (!seen_sof)
            bad("no frame header");
        if (progressive)
            finish_progressive();
    }

    // ---- planes -> RGBA8
    static uint8_t div4(int x) { return (uint8_t)(x >> 2); }
    static uint8_t div16(int x) { return (uint8_t)(x >> 4); }
    static const uint8_t *resample(uint8_t *out, const uint8_t *near_, const uint8_t *far_, int w, int hs, int vs) {
        if (hs == 1 && vs == 1)
            return near_;
        if (hs == 1 && vs == 2) {
            for (int i = 0; i < w; ++i)
                out[i] = div4(3 * near_[i] + far_[i] + 2);
            return out;
        }
        if (hs == 2 && vs == 1) {
            if (w == 1) {
                out[0] = out[1] = near_[0];
                return out;
            }
            out[0] = near_[0];
            out[1] = div4(near_[0] * 3 + near_[1] + 2);
            int i;
            for (i = 1; i < w - 1; ++i) {
                const int n = 3 * near_[i] + 2;
                out[i * 2] = div4(n + near_[i - 1]);
                out[i * 2 + 1] = div4(n + near_[i + 1]);
            }
            out[i * 2] = div4(near_[w - 2] * 3 + near_[w - 1] + 2);
            out[i * 2 + 1] = near_[w - 1];
            return out;
        }
        if (hs == 2 && vs == 2) {
            if (w == 1) {
                out[0] = out[1] = div4(3 * near_[0] + far_[0] + 2);
                return out;
            }
            int t1 = 3 * near_[0] + far_[0];
            out[0] = div4(t1 + 2);
            for (int i = 1; i < w; ++i) {
                const int t0 = t1;
                t1 = 3 * near_[i] + far_[i];
                out[i * 2 - 1] = div16(3 * t0 + t1 + 8);
                out[i * 2] = div16(3 * t1 + t0 + 8);
            }
            out[w * 2 - 1] = div4(t1 + 2);
            return out;
        }
        for (int i = 0; i < w; ++i) // any other ratio: replicate
            for (int j = 0; j < hs; ++j)
                out[i * hs + j] = near_[i];
        return out;
    }

    uint8_t *to_rgba() {
        uint8_t *out = (uint8_t *)std::malloc((size_t)width * height * 4);
        if (!out)
            throw std::bad_alloc();
        int rgb_ids = 0;
        for (int i = 0; i < ncomp; ++i)
            if (comp[i].id == "RGB"[i % 3])
                ++rgb_ids;
        const bool is_rgb = ncomp == 3 && (rgb_ids == 3 || (adobe_transform == 0 && !jfif));
        struct Res {
            int hs, vs, ystep, w_lores, ypos;
            const uint8_t *line0, *line1;
            std::vector<uint8_t> buf;
        } res[4];
        for (int k = 0; k < ncomp; ++k) {
            Res &r = res[k];
            r.hs = hmax / comp[k].h;
            r.vs = vmax / comp[k].v;
            r.ystep = r.vs >> 1;
            r.w_lores = (width + r.hs - 1) / r.hs;
            r.ypos = 0;
            r.line0 = r.line1 = comp[k].plane.data();
            r.buf.assign((size_t)width + 3 + 8, 0);
        }
        const int K_cr_r = fx(1.40200f) << 8, K_cr_g = fx(0.71414f) << 8, K_cb_g = fx(0.34414f) << 8, K_cb_b = fx(1.77200f) << 8;
        for (int j = 0; j < height; ++j) {
            const uint8_t *row[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int k = 0; k < ncomp; ++k) {
                Res &r = res[k];
                const bool y_bot = r.ystep >= (r.vs >> 1);
                row[k] = resample(r.buf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[k].y)
                        r.line1 += comp[k].w2;
                }
            }
            uint8_t *o = out + (size_t)j * width * 4;
            if (ncomp == 1) {
                for (int i = 0; i < width; ++i, o += 4) {
                    o[0] = o[1] = o[2] = row[0][i];
                    o[3] = 255;
                }
            } else if (is_rgb) {
                for (int i = 0; i < width; ++i, o += 4) {
                    o[0] = row[0][i];
                    o[1] = row[1][i];
                    o[2] = row[2][i];
                    o[3] = 255;
                }
            } else {
                for (int i = 0; i < width; ++i, o += 4) {
                    const int y_fixed = (row[0][i] << 20) + (1 << 19);
                    const int cr = row[2][i] - 128, cb = row[1][i] - 128;
                    int r = y_fixed + cr * K_cr_r;
                    int g = y_fixed + cr * -K_cr_g + (int)((uint32_t)(cb * -K_cb_g) & 0xffff0000u);
                    int b = y_fixed + cb * K_cb_b;
                    r >>= 20;
                    g >>= 20;
                    b >>= 20;
                    o[0] = clamp8(r);
                    o[1] = clamp8(g);
                    o[2] = clamp8(b);
                    o[3] = 255;
                }
            }
        }
        return out;
    }
};

std::vector<uint8_t> read_file(const char *path) {
    std::vector<uint8_t> file;
    FILE *f = std::fopen(path, "rb");
    if (!f)
        return file;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0)
        file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    return file;
}

} // namespace

extern "C" int rt_jpeg_decode_file(const char *path, uint32_t *w_out, uint32_t *h_out, uint8_t **rgba_out) {
    if (!path || !w_out || !h_out || !rgba_out)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_jpeg_decode_file: null argument");
    const std::vector<uint8_t> file = read_file(path);
    if (file.empty())
        return rt::fail(RT_ERR_IO, std::string("Failed to load image from ") + path); // geometry.h:588
    try {
        Decoder d;
        d.data = file.data();
        d.end = file.data() + file.size();
        d.parse();
        *rgba_out = d.to_rgba();
        *w_out = (uint32_t)d.width;
        *h_out = (uint32_t)d.height;
    } catch (const JpegError &e) {
        return rt::fail(RT_ERR_FORMAT, std::string(path) + ": " + e.msg);
    } catch (const std::bad_alloc &) {
        return rt::fail(RT_ERR_OOM, "JPEG: out of memory");
    }
    return RT_OK;
}

// Texture::load_img (geometry.h:584-598) for the formats this loader reads: PNG, JPEG and Radiance HDR (the environment map's default
// format, hdr_decode.cpp), told apart by their signatures.
extern "C" int rt_image_decode_file(const char *path, uint32_t *w_out, uint32_t *h_out, uint8_t **rgba_out) {
    if (!path)
        return rt::fail(RT_ERR_INVALID_ARG, "rt_image_decode_file: null argument");
    uint8_t sig[3] = {0, 0, 0};
    if (FILE *f = std::fopen(path, "rb")) {
        const size_t n = std::fread(sig, 1, 3, f);
        (void)n;
        std::fclose(f);
    } else {
        return rt::fail(RT_ERR_IO, std::string("Failed to load image from ") + path);
    }
    if (sig[0] == 0xFF && sig[1] == 0xD8 && sig[2] == 0xFF)
        return rt_jpeg_decode_file(path, w_out, h_out, rgba_out);
    if (sig[0] == '#' && sig[1] == '?')
        return rt_hdr_decode_file(path, w_out, h_out, rgba_out);
    return rt_png_decode_file(path, w_out, h_out, rgba_out);
}

#!/usr/bin/env python3
"""JPEG fixtures for the host texture decoder (csrc/host/jpeg_decode.cpp), container only.

A small JPEG ENCODER (ITU-T T.81: baseline and progressive Huffman, written here so that no image library is needed) produces
files that cover what a decoder has to get right — sampling factors 4:4:4 / 4:2:2 / 4:4:0 / 4:2:0 / 4:1:1, grey, RGB component
ids, odd sizes, restart intervals, non-interleaved scans, 16-bit quantisation tables, progressive scans with spectral selection
and successive approximation — and the REFERENCE's own image decoder (stb_image v2.30 as compiled into oracle/_ref/ref_probe,
geometry::Texture::load_img) says what texels each of them decodes to: tests/golden/jpeg/expected.npz.
Only data is stored: the JPEG inputs and the reference's outputs.   python tests/golden/make_jpeg_golden.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402

OUT = os.path.join(HERE, "jpeg")

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
          29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
# T.81 Annex K: example quantisation tables (natural order) and Huffman tables
Q_LUM = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
         18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99]
Q_CHR = [17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32
DC_LUM = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_CHR = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_LUM = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7D],
          [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xA1, 0x08, 0x23, 0x42, 0xB1, 0xC1,
           0x15, 0x52, 0xD1, 0xF0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0A, 0x16, 0x17, 0x18, 0x19, 0x1A, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39,
           0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74, 0x75,
           0x76, 0x77, 0x78, 0x79, 0x7A, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3, 0xA4, 0xA5, 0xA6, 0xA7,
           0xA8, 0xA9, 0xAA, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3, 0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6, 0xD7, 0xD8,
           0xD9, 0xDA, 0xE1, 0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF1, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9, 0xFA])
AC_CHR = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
          [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xA1, 0xB1, 0xC1, 0x09,
           0x23, 0x33, 0x52, 0xF0, 0x15, 0x62, 0x72, 0xD1, 0x0A, 0x16, 0x24, 0x34, 0xE1, 0x25, 0xF1, 0x17, 0x18, 0x19, 0x1A, 0x26, 0x27, 0x28, 0x29, 0x2A, 0x35, 0x36, 0x37, 0x38,
           0x39, 0x3A, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4A, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5A, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6A, 0x73, 0x74,
           0x75, 0x76, 0x77, 0x78, 0x79, 0x7A, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8A, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9A, 0xA2, 0xA3, 0xA4, 0xA5,
           0xA6, 0xA7, 0xA8, 0xA9, 0xAA, 0xB2, 0xB3, 0xB4, 0xB5, 0xB6, 0xB7, 0xB8, 0xB9, 0xBA, 0xC2, 0xC3, 0xC4, 0xC5, 0xC6, 0xC7, 0xC8, 0xC9, 0xCA, 0xD2, 0xD3, 0xD4, 0xD5, 0xD6,
           0xD7, 0xD8, 0xD9, 0xDA, 0xE2, 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xEA, 0xF2, 0xF3, 0xF4, 0xF5, 0xF6, 0xF7, 0xF8, 0xF9, 0xFA])
# progressive AC scans need EOBn symbols the example tables lack: a table with all 256 symbols (254 codes of 8 bits, 2 of 9)
AC_ALL = ([0, 0, 0, 0, 0, 0, 0, 254, 2, 0, 0, 0, 0, 0, 0, 0], list(range(256)))


def huff_codes(table):
    counts, symbols = table
    codes, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            codes[symbols[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return codes


class BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, nbits):
        if nbits == 0:
            return
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1))
        self.n += nbits
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)

    def marker(self, m):
        self.flush()
        self.out += bytes([0xFF, m])


def nbits_of(v):
    return int(v).bit_length()


DCT = np.array([[(np.sqrt(1 / 8) if u == 0 else np.sqrt(2 / 8)) * np.cos((2 * x + 1) * u * np.pi / 16) for x in range(8)] for u in range(8)])


def blocks_of(plane, quant):
    """plane (h2, w2) uint8 -> quantised coefficients (bh, bw, 64) int, natural order."""
    h2, w2 = plane.shape
    p = plane.astype(np.float64) - 128.0
    b = p.reshape(h2 // 8, 8, w2 // 8, 8).transpose(0, 2, 1, 3)
    c = np.einsum("ux,abxy,vy->abuv", DCT, b, DCT)
    q = np.asarray(quant, dtype=np.float64).reshape(8, 8)
    return np.round(c / q).astype(np.int64).reshape(h2 // 8, w2 // 8, 64)


def seg(m, payload):
    return bytes([0xFF, m]) + struct.pack(">H", len(payload) + 2) + payload


def dqt(tid, table, sixteen=False):
    zz = [table[ZIGZAG[i]] for i in range(64)]
    if sixteen:
        return seg(0xDB, bytes([0x10 | tid]) + b"".join(struct.pack(">H", v) for v in zz))
    return seg(0xDB, bytes([tid]) + bytes(zz))


def dht(tc, th, table):
    return seg(0xC4, bytes([(tc << 4) | th]) + bytes(table[0]) + bytes(table[1]))


class Image:
    """Component planes padded to whole MCUs + their quantised coefficients."""

    def __init__(self, rgb, sampling, quality_scale, color="ycc"):
        h, w, _ = rgb.shape
        self.w, self.h = w, h
        f = rgb.astype(np.float64)
        if color == "grey":
            planes = [0.299 * f[..., 0] + 0.587 * f[..., 1] + 0.114 * f[..., 2]]
        elif color == "rgb":
            planes = [f[..., 0], f[..., 1], f[..., 2]]
        else:
            y = 0.299 * f[..., 0] + 0.587 * f[..., 1] + 0.114 * f[..., 2]
            planes = [y, 128 - 0.168736 * f[..., 0] - 0.331264 * f[..., 1] + 0.5 * f[..., 2], 128 + 0.5 * f[..., 0] - 0.418688 * f[..., 1] - 0.081312 * f[..., 2]]
        self.sampling = sampling[: len(planes)]
        self.hmax = max(s[0] for s in self.sampling)
        self.vmax = max(s[1] for s in self.sampling)
        self.mcus_x = -(-w // (8 * self.hmax))
        self.mcus_y = -(-h // (8 * self.vmax))
        self.quant = [[max(1, min(255, int(round(v * quality_scale)))) for v in (Q_LUM if i == 0 or color == "rgb" else Q_CHR)] for i in range(len(planes))]
        self.coef = []
        self.size = []
        for i, (p, (hs, vs)) in enumerate(zip(planes, self.sampling)):
            fx, fy = self.hmax // hs, self.vmax // vs
            cw, ch = -(-w * hs // self.hmax), -(-h * vs // self.vmax)
            pp = np.pad(p, ((0, ch * fy - h), (0, cw * fx - w)), mode="edge")
            sub = pp.reshape(ch, fy, cw, fx).mean(axis=(1, 3))  # box filter
            w2, h2 = self.mcus_x * hs * 8, self.mcus_y * vs * 8
            sub = np.pad(sub, ((0, h2 - ch), (0, w2 - cw)), mode="edge")
            self.size.append((cw, ch))
            self.coef.append(blocks_of(np.clip(np.round(sub), 0, 255).astype(np.uint8), self.quant[i]))


def frame_segments(img, sof, ids, jfif=True, adobe=None, sixteen_bit_q=False):
    out = b"\xFF\xD8"
    if jfif:
        out += seg(0xE0, b"JFIF\0\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    if adobe is not None:
        out += seg(0xEE, b"Adobe\0" + bytes([100, 0, 0, 0, 0, adobe]))
    out += seg(0xFE, b"fixture written by make_jpeg_golden.py")
    nq = len(img.coef)
    for i in range(nq):
        out += dqt(i, img.quant[i], sixteen=sixteen_bit_q and i == 0)
    comps = b"".join(bytes([ids[i], (img.sampling[i][0] << 4) | img.sampling[i][1], i]) for i in range(nq))
    out += seg(sof, bytes([8]) + struct.pack(">HH", img.h, img.w) + bytes([nq]) + comps)
    return out


def encode_baseline(img, ids=(1, 2, 3), restart=0, interleaved=True, **kw):
    n = len(img.coef)
    out = frame_segments(img, 0xC0, ids, **kw)
    out += dht(0, 0, DC_LUM) + dht(1, 0, AC_LUM)
    if n > 1:
        out += dht(0, 1, DC_CHR) + dht(1, 1, AC_CHR)
    if restart:
        out += seg(0xDD, struct.pack(">H", restart))
    dcc = [huff_codes(DC_LUM), huff_codes(DC_CHR)]
    acc = [huff_codes(AC_LUM), huff_codes(AC_CHR)]

    def block(bw, comp, c, pred):
        t = 0 if comp == 0 or ids[0] == ord("R") else 1
        diff = int(c[0]) - pred
        s = nbits_of(abs(diff))
        bw.put(*dcc[t][s])
        if s:
            bw.put(diff if diff >= 0 else diff + (1 << s) - 1, s)
        run = 0
        for k in range(1, 64):
            v = int(c[ZIGZAG[k]])
            if v == 0:
                run += 1
                continue
            while run > 15:
                bw.put(*acc[t][0xF0])
                run -= 16
            s = nbits_of(abs(v))
            bw.put(*acc[t][(run << 4) | s])
            bw.put(v if v >= 0 else v + (1 << s) - 1, s)
            run = 0
        if run:
            bw.put(*acc[t][0x00])
        return int(c[0])

    def table_sel(comp):
        return 0x00 if comp == 0 or ids[0] == ord("R") else 0x11

    scans = [list(range(n))] if interleaved or n == 1 else [[i] for i in range(n)]
    for comps in scans:
        out += seg(0xDA, bytes([len(comps)]) + b"".join(bytes([ids[c], table_sel(c)]) for c in comps) + bytes([0, 63, 0]))
        bw = BitWriter()
        pred = [0] * n
        count, rst = 0, 0
        if len(comps) == 1 and n > 1 or (n == 1):
            c = comps[0]
            cw, ch = img.size[c]
            units = [(c, bx, by) for by in range(-(-ch // 8)) for bx in range(-(-cw // 8))]
            groups = [[u] for u in units]
        else:
            groups = []
            for my in range(img.mcus_y):
                for mx in range(img.mcus_x):
                    g = []
                    for c in comps:
                        hs, vs = img.sampling[c]
                        g += [(c, mx * hs + hx, my * vs + vy) for vy in range(vs) for hx in range(hs)]
                    groups.append(g)
        for gi, g in enumerate(groups):
            for (c, bx, by) in g:
                pred[c] = block(bw, c, img.coef[c][by, bx], pred[c])
            count += 1
            if restart and count == restart and gi != len(groups) - 1:
                bw.marker(0xD0 + (rst & 7))
                rst += 1
                count = 0
                pred = [0] * n
        bw.flush()
        out += bytes(bw.out)
    return out + b"\xFF\xD9"


def encode_progressive(img, script, ids=(1, 2, 3), restart=0, **kw):
    """script: list of (components, Ss, Se, Ah, Al)."""
    n = len(img.coef)
    out = frame_segments(img, 0xC2, ids, **kw)
    out += dht(0, 0, DC_LUM) + dht(0, 1, DC_CHR) + dht(1, 0, AC_ALL)
    if restart:
        out += seg(0xDD, struct.pack(">H", restart))
    dcc = [huff_codes(DC_LUM), huff_codes(DC_CHR)]
    acc = huff_codes(AC_ALL)
    for (comps, ss, se, ah, al) in script:
        sel = b"".join(bytes([ids[c], (0x00 if c == 0 else 0x10)]) for c in comps)
        out += seg(0xDA, bytes([len(comps)]) + sel + bytes([ss, se, (ah << 4) | al]))
        bw = BitWriter()
        state = {"eobrun": 0, "be": []}

        def emit_eobrun():
            if state["eobrun"]:
                nb = nbits_of(state["eobrun"]) - 1
                bw.put(*acc[nb << 4])
                if nb:
                    bw.put(state["eobrun"] & ((1 << nb) - 1), nb)
                state["eobrun"] = 0
            for b in state["be"]:
                bw.put(b, 1)
            state["be"] = []

        if ss == 0:  # DC scan (may interleave)
            if len(comps) == 1:
                c = comps[0]
                cw, ch = img.size[c]
                groups = [[(c, bx, by)] for by in range(-(-ch // 8)) for bx in range(-(-cw // 8))]
            else:
                groups = []
                for my in range(img.mcus_y):
                    for mx in range(img.mcus_x):
                        g = []
                        for c in comps:
                            hs, vs = img.sampling[c]
                            g += [(c, mx * hs + hx, my * vs + vy) for vy in range(vs) for hx in range(hs)]
                        groups.append(g)
            pred = [0] * n
            count, rst = 0, 0
            for gi, g in enumerate(groups):
                for (c, bx, by) in g:
                    v = int(img.coef[c][by, bx][0])
                    if ah == 0:
                        v >>= al  # arithmetic shift (point transform of the DC coefficient)
                        diff = v - pred[c]
                        pred[c] = v
                        s = nbits_of(abs(diff))
                        bw.put(*dcc[0 if c == 0 else 1][s])
                        if s:
                            bw.put(diff if diff >= 0 else diff + (1 << s) - 1, s)
                    else:
                        bw.put((v >> al) & 1, 1)
                count += 1
                if restart and count == restart and gi != len(groups) - 1:
                    bw.marker(0xD0 + (rst & 7))
                    rst += 1
                    count = 0
                    pred = [0] * n
        else:  # AC scan of one component
            c = comps[0]
            cw, ch = img.size[c]
            blocks = [(bx, by) for by in range(-(-ch // 8)) for bx in range(-(-cw // 8))]
            count, rst = 0, 0
            for bi, (bx, by) in enumerate(blocks):
                co = img.coef[c][by, bx]
                if ah == 0:
                    run = 0
                    for k in range(ss, se + 1):
                        v = int(co[ZIGZAG[k]])
                        a = abs(v) >> al
                        if a == 0:
                            run += 1
                            continue
                        emit_eobrun()
                        while run > 15:
                            bw.put(*acc[0xF0])
                            run -= 16
                        s = nbits_of(a)
                        bw.put(*acc[(run << 4) | s])
                        bw.put(a if v >= 0 else (~a) & ((1 << s) - 1), s)
                        run = 0
                    if run:
                        state["eobrun"] += 1
                        if state["eobrun"] == 0x7FFF:
                            emit_eobrun()
                else:
                    absv = [abs(int(co[ZIGZAG[k]])) >> al for k in range(64)]
                    eob = 0
                    for k in range(ss, se + 1):
                        if absv[k] == 1:
                            eob = k
                    run, br = 0, []
                    for k in range(ss, se + 1):
                        t = absv[k]
                        if t == 0:
                            run += 1
                            continue
                        while run > 15 and k <= eob:
                            emit_eobrun()
                            bw.put(*acc[0xF0])
                            run -= 16
                            for b in br:
                                bw.put(b, 1)
                            br = []
                        if t > 1:
                            br.append(t & 1)
                            continue
                        emit_eobrun()
                        bw.put(*acc[(run << 4) | 1])
                        bw.put(0 if int(co[ZIGZAG[k]]) < 0 else 1, 1)
                        for b in br:
                            bw.put(b, 1)
                        br = []
                        run = 0
                    if run or br:
                        state["eobrun"] += 1
                        state["be"] += br
                        if state["eobrun"] == 0x7FFF or len(state["be"]) > 900:
                            emit_eobrun()
                count += 1
                if restart and count == restart and bi != len(blocks) - 1:
                    emit_eobrun()
                    bw.marker(0xD0 + (rst & 7))
                    rst += 1
                    count = 0
            emit_eobrun()
        bw.flush()
        out += bytes(bw.out)
    return out + b"\xFF\xD9"


def test_image(rng, w, h):
    """Smooth gradients + blobs + a few sharp edges: every block gets low and high frequencies."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        img[..., c] = 128 + 80 * np.sin(xx / (5 + 3 * c) + c) * np.cos(yy / (7 - c)) + 30 * rng.standard_normal((h, w))
    img[h // 3 : h // 3 + 3, :, 0] = 255
    img[:, w // 2 : w // 2 + 2, 1] = 0
    img[(xx + yy) % 11 < 2] += 40
    return np.clip(np.round(img), 0, 255).astype(np.uint8)


PROG_FULL = [((0, 1, 2), 0, 0, 0, 1), ((0,), 1, 5, 0, 2), ((2,), 1, 63, 0, 1), ((1,), 1, 63, 0, 1), ((0,), 6, 63, 0, 2), ((0,), 1, 63, 2, 1),
             ((0, 1, 2), 0, 0, 1, 0), ((2,), 1, 63, 1, 0), ((1,), 1, 63, 1, 0), ((0,), 1, 63, 1, 0)]
PROG_SPECTRAL = [((0, 1, 2), 0, 0, 0, 0), ((0,), 1, 9, 0, 0), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0), ((0,), 10, 63, 0, 0)]
PROG_GREY = [((0,), 0, 0, 0, 2), ((0,), 1, 63, 0, 1), ((0,), 0, 0, 2, 1), ((0,), 0, 0, 1, 0), ((0,), 1, 63, 1, 0)]


def main():
    assert oracle.have_reference_build()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(77)
    S444, S422, S440, S420, S411 = [(1, 1)] * 3, [(2, 1), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)], [(2, 2), (1, 1), (1, 1)], [(4, 1), (1, 1), (1, 1)]
    files = {}
    for name, samp, (w, h) in (("444", S444, (45, 37)), ("422", S422, (45, 37)), ("440", S440, (37, 45)), ("420", S420, (61, 43)), ("411", S411, (70, 19)),
                               ("420_tiny", S420, (1, 1)), ("420_thin", S420, (3, 50)), ("420_wide", S420, (50, 3)), ("422_w1", S422, (1, 9)), ("420_16x16", S420, (16, 16))):
        files["base_" + name] = encode_baseline(Image(test_image(rng, w, h), samp, 0.5))
    files["base_420_q_coarse"] = encode_baseline(Image(test_image(rng, 40, 40), S420, 3.0))
    files["base_420_q_fine"] = encode_baseline(Image(test_image(rng, 40, 40), S420, 0.1))
    files["base_420_restart3"] = encode_baseline(Image(test_image(rng, 70, 50), S420, 0.6), restart=3)
    files["base_444_restart1"] = encode_baseline(Image(test_image(rng, 33, 17), S444, 0.6), restart=1)
    files["base_420_noninterleaved"] = encode_baseline(Image(test_image(rng, 53, 39), S420, 0.5), interleaved=False)
    files["base_420_noninterleaved_restart"] = encode_baseline(Image(test_image(rng, 53, 39), S420, 0.5), interleaved=False, restart=4)
    files["base_grey"] = encode_baseline(Image(test_image(rng, 41, 29), S444, 0.5, color="grey"))
    files["base_grey_restart"] = encode_baseline(Image(test_image(rng, 41, 29), S444, 0.5, color="grey"), restart=5)
    files["base_rgb_ids"] = encode_baseline(Image(test_image(rng, 30, 22), S444, 0.4, color="rgb"), ids=(ord("R"), ord("G"), ord("B")))
    files["base_adobe_rgb_nojfif"] = encode_baseline(Image(test_image(rng, 30, 22), S444, 0.4, color="rgb"), jfif=False, adobe=0)
    files["base_adobe_ycc_nojfif"] = encode_baseline(Image(test_image(rng, 30, 22), S420, 0.4), jfif=False, adobe=1)
    files["base_nojfif_noadobe"] = encode_baseline(Image(test_image(rng, 30, 22), S444, 0.4), jfif=False)
    files["base_adobe0_with_jfif"] = encode_baseline(Image(test_image(rng, 30, 22), S444, 0.4), jfif=True, adobe=0)
    files["base_16bit_dqt"] = encode_baseline(Image(test_image(rng, 34, 26), S420, 0.5), sixteen_bit_q=True)
    files["prog_420_full"] = encode_progressive(Image(test_image(rng, 61, 43), S420, 0.4), PROG_FULL)
    files["prog_444_full"] = encode_progressive(Image(test_image(rng, 45, 37), S444, 0.3), PROG_FULL)
    files["prog_422_spectral"] = encode_progressive(Image(test_image(rng, 45, 37), S422, 0.5), PROG_SPECTRAL)
    files["prog_420_full_restart"] = encode_progressive(Image(test_image(rng, 70, 50), S420, 0.4), PROG_FULL, restart=2)
    files["prog_grey"] = encode_progressive(Image(test_image(rng, 41, 29), S444, 0.3, color="grey"), PROG_GREY)
    files["prog_420_fine"] = encode_progressive(Image(test_image(rng, 48, 48), S420, 0.08), PROG_FULL)
    expected = {}
    with tempfile.TemporaryDirectory() as td:
        for name, blob in files.items():
            path = os.path.join(OUT, name + ".jpg")
            with open(path, "wb") as f:
                f.write(blob)
            out = os.path.join(td, name + ".bin")
            subprocess.check_call([oracle.REF_PROBE, "texture", path, "0", "0", out])
            words = np.fromfile(out, dtype=np.uint32)
            w, h = int(words[0]), int(words[1])
            texels = words[2:].view(np.float32).reshape(h, w, 4)
            u8 = np.round(texels * 255.0).astype(np.uint8)
            assert np.array_equal((u8 / np.float32(255.0)).astype(np.float32), texels)
            expected[name] = u8
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)
    print(len(files), "JPEG fixtures,", sum(len(b) for b in files.values()), "bytes")


if __name__ == "__main__":
    main()

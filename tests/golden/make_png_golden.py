#!/usr/bin/env python3
"""PNG fixtures for the host texture decoder (csrc/host/png_decode.cpp), container only.

Writes small PNG files covering every colour type / bit depth / interlace combination PNG allows, with and without tRNS,
under tests/golden/png/, and asks the REFERENCE's own image decoder (stb_image v2.30 as compiled into oracle/_ref/ref_probe,
geometry::Texture::load_img, 4 channels forced) what texels it returns for each: tests/golden/png/expected.npz.
Only data is stored: the PNG inputs (written by the encoder below) and the reference's outputs.   python tests/golden/make_png_golden.py
"""
import os
import struct
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle  # noqa: E402

OUT = os.path.join(HERE, "png")
CH = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def pack_rows(samples, depth, filters, salt):
    """samples: (h, w, ch) integer samples of `depth` bits -> filtered scanlines (filter type cycles through `filters`)."""
    h, w, ch = samples.shape
    if depth == 16:
        rows = samples.astype(">u2").reshape(h, -1).view(np.uint8).reshape(h, -1)
        bpp = 2 * ch
    elif depth == 8:
        rows = samples.astype(np.uint8).reshape(h, -1)
        bpp = ch
    else:
        per = 8 // depth
        flat = samples.reshape(h, -1).astype(np.uint32)
        pad = (-flat.shape[1]) % per
        flat = np.pad(flat, ((0, 0), (0, pad)))
        rows = np.zeros((h, flat.shape[1] // per), dtype=np.uint32)
        for k in range(per):
            rows |= flat[:, k::per] << (8 - depth - k * depth)
        rows = rows.astype(np.uint8)
        bpp = 1
    out = bytearray()
    prev = np.zeros(rows.shape[1], dtype=np.int32)
    for y in range(h):
        row = rows[y].astype(np.int32)
        ft = filters[(y + salt) % len(filters)]
        left = np.concatenate([np.zeros(bpp, np.int32), row[:-bpp]]) if rows.shape[1] > bpp else np.zeros_like(row)
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if rows.shape[1] > bpp else np.zeros_like(row)
        if ft == 0:
            enc = row
        elif ft == 1:
            enc = row - left
        elif ft == 2:
            enc = row - prev
        elif ft == 3:
            enc = row - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            enc = row - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        out += bytes([ft]) + (enc & 255).astype(np.uint8).tobytes()
        prev = row
    return bytes(out)


def write_png(path, samples, ctype, depth, interlace=False, plte=None, trns=None, filters=(0, 1, 2, 3, 4), extra=b""):
    h, w, _ = samples.shape
    if interlace:
        raw = b""
        for k, (xs, ys, dx, dy) in enumerate([(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]):
            sub = samples[ys::dy, xs::dx]
            if sub.size:
                raw += pack_rows(sub, depth, filters, k)
    else:
        raw = pack_rows(samples, depth, filters, 0)
    comp = zlib.compress(raw, 9)
    cut = max(1, len(comp) // 3)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0)))
        f.write(extra)
        if plte is not None:
            f.write(chunk(b"PLTE", bytes(plte)))
        if trns is not None:
            f.write(chunk(b"tRNS", bytes(trns)))
        f.write(chunk(b"IDAT", comp[:cut]) + chunk(b"IDAT", comp[cut:]) + chunk(b"IEND", b""))


def main():
    assert oracle.have_reference_build()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(2024)
    cases = []
    for ctype, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        for depth in depths:
            for interlace in (False, True):
                for with_trns in ((False, True) if ctype in (0, 2, 3) else (False,)):
                    w, h = (13, 9) if not interlace else (11, 10)
                    if depth < 8 and interlace:
                        w, h = 19, 7  # rows that end in the middle of a byte in several passes
                    hi = (1 << depth) - 1
                    samples = rng.integers(0, hi + 1, size=(h, w, CH[ctype]), dtype=np.uint32)
                    plte = trns = None
                    if ctype == 3:
                        n_pal = min(1 << depth, 200)
                        plte = rng.integers(0, 256, size=n_pal * 3, dtype=np.uint8)
                        samples = rng.integers(0, n_pal, size=(h, w, 1), dtype=np.uint32)
                        if with_trns:
                            trns = rng.integers(0, 256, size=max(1, n_pal // 2), dtype=np.uint8)
                    elif with_trns:
                        key = samples[h // 2, w // 2].copy()  # a colour that occurs (several times for low depths)
                        samples[0, 0] = key
                        samples[-1, -1] = key
                        trns = b"".join(struct.pack(">H", int(v)) for v in key)
                    name = f"c{ctype}_d{depth}{'_i' if interlace else ''}{'_t' if with_trns else ''}"
                    extra = chunk(b"gAMA", struct.pack(">I", 45455)) + chunk(b"tEXt", b"Comment\x00fixture") if (depth == 8 and not interlace) else b""
                    write_png(os.path.join(OUT, name + ".png"), samples, ctype, depth, interlace, plte, trns, extra=extra)
                    cases.append(name)
    # a 1x1 image and a wide single row, each colour type at 8 bits
    for ctype in (0, 2, 3, 4, 6):
        for name, (w, h) in (("1x1", (1, 1)), ("row", (37, 1))):
            plte = rng.integers(0, 256, size=16 * 3, dtype=np.uint8) if ctype == 3 else None
            samples = rng.integers(0, 16 if ctype == 3 else 256, size=(h, w, CH[ctype]), dtype=np.uint32)
            nm = f"c{ctype}_{name}"
            write_png(os.path.join(OUT, nm + ".png"), samples, ctype, 8, interlace=(name == "row"), plte=plte)
            cases.append(nm)
    expected = {}
    with tempfile.TemporaryDirectory() as td:
        for name in cases:
            out = os.path.join(td, name + ".bin")
            subprocess.check_call([oracle.REF_PROBE, "texture", os.path.join(OUT, name + ".png"), "0", "0", out])
            words = np.fromfile(out, dtype=np.uint32)
            w, h = int(words[0]), int(words[1])
            texels = words[2:].view(np.float32).reshape(h, w, 4)
            u8 = np.round(texels * 255.0).astype(np.uint8)
            assert np.array_equal((u8 / np.float32(255.0)).astype(np.float32), texels)  # texels are exactly k / 255.0f
            expected[name] = u8
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)
    print(len(cases), "PNG fixtures,", sum(os.path.getsize(os.path.join(OUT, c + ".png")) for c in cases), "bytes")


if __name__ == "__main__":
    main()

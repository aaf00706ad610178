"""The host PNG decoder (csrc/host/png_decode.cpp) against what the REFERENCE's stb_image build returns for the same files
(tests/golden/png/*.png + expected.npz, made by tests/golden/make_png_golden.py through oracle/_ref/ref_probe): every colour
type x bit depth x interlace combination of the PNG format, with and without tRNS, ancillary chunks, 1x1 and one-row images.
CPU only."""
import os

import numpy as np
import pytest

PNG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png")
EXPECTED = np.load(os.path.join(PNG, "expected.npz"))


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
def test_png_decoder_equals_the_reference_stb_image(rt, name):
    got = rt.png_decode(os.path.join(PNG, name + ".png"))
    want = EXPECTED[name]
    assert got.shape == want.shape and np.array_equal(got, want), name


def test_fixture_set_covers_the_format():
    names = set(EXPECTED.files)
    for ctype, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        for d in depths:
            assert {f"c{ctype}_d{d}", f"c{ctype}_d{d}_i"} <= names
    assert {"c0_d16_t", "c2_d16_i_t", "c3_d2_t", "c0_d1_i_t"} <= names and len(names) >= 50


def test_other_formats_are_named_in_the_error(rt, tmp_path):
    """stb_image would also read BMP / GIF / ...; this loader does not, and says which format it met (JPEG goes to the JPEG
    decoder through rt_image_decode_file; the PNG entry point alone names it too)."""
    for blob, word in ((b"\xff\xd8\xff\xe0" + b"\x00" * 32, "JPEG"), (b"BM" + b"\x00" * 32, "BMP"), (b"GIF89a" + b"\x00" * 32, "GIF")):
        p = tmp_path / "x.bin"
        p.write_bytes(blob)
        with pytest.raises(rt.RtError) as e:
            rt.png_decode(str(p))
        assert e.value.code == 6 and word in str(e.value) and "PNG" in str(e.value)

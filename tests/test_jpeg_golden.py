"""The host JPEG decoder (csrc/host/jpeg_decode.cpp) against what the REFERENCE's stb_image build returns for the same files
(tests/golden/jpeg/*.jpg + expected.npz, made by tests/golden/make_jpeg_golden.py through oracle/_ref/ref_probe): baseline and
progressive, every common chroma sampling, grey, RGB by component ids / Adobe marker, odd and tiny sizes, restart intervals,
non-interleaved scans, 16-bit quantisation tables, coarse and fine quantisation. Byte for byte. CPU only."""
import os

import numpy as np
import pytest

JPG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
EXPECTED = np.load(os.path.join(JPG, "expected.npz"))


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
def test_jpeg_decoder_equals_the_reference_stb_image(rt, name):
    got = rt.image_decode(os.path.join(JPG, name + ".jpg"))
    want = EXPECTED[name]
    assert got.shape == want.shape, (got.shape, want.shape)
    diff = int((got != want).sum())
    assert diff == 0, f"{name}: {diff} of {want.size} bytes differ, max |d| {int(np.abs(got.astype(int) - want.astype(int)).max())}"
    assert (got[..., 3] == 255).all()


def test_fixture_set_covers_the_format():
    names = set(EXPECTED.files)
    assert {"base_444", "base_422", "base_440", "base_420", "base_411", "base_grey", "base_rgb_ids", "base_420_restart3", "base_420_noninterleaved",
            "base_16bit_dqt", "prog_420_full", "prog_444_full", "prog_422_spectral", "prog_420_full_restart", "prog_grey"} <= names and len(names) >= 30
    # the fixtures are real images, not flat fields: the decoder's arithmetic is exercised
    assert EXPECTED["base_420"].std() > 30 and EXPECTED["prog_420_full"].std() > 30


def test_unsupported_and_malformed_jpeg_are_errors(rt, tmp_path):
    good = open(os.path.join(JPG, "base_420.jpg"), "rb").read()
    cases = {
        "truncated_header": good[:30],
        "no_sof": good[:2] + good[good.index(b"\xff\xda") :],
        "arithmetic": good.replace(b"\xff\xc0", b"\xff\xc9", 1),
        "lossless": good.replace(b"\xff\xc0", b"\xff\xc3", 1),
        "twelve_bit": good.replace(b"\xff\xc0\x00\x11\x08", b"\xff\xc0\x00\x11\x0c", 1),
    }
    for name, blob in cases.items():
        p = tmp_path / (name + ".jpg")
        p.write_bytes(blob)
        with pytest.raises(rt.RtError) as e:
            rt.image_decode(str(p))
        assert e.value.code == 6, name  # RT_ERR_FORMAT
    # a scan cut short is not an error in stb_image either (missing data decodes as zero bits): it must not crash
    p = tmp_path / "cut.jpg"
    p.write_bytes(good[: len(good) * 2 // 3])
    out = rt.image_decode(str(p))
    assert out.shape == EXPECTED["base_420"].shape
    with pytest.raises(rt.RtError) as e:
        rt.image_decode(str(tmp_path / "missing.jpg"))
    assert e.value.code == 5


def test_gltf_with_jpeg_textures_matches_the_reference_binary(rt, sg, oracle, tmp_path):
    """A glTF whose images are .jpg files (what real assets such as Khronos Sponza ship): the loader hands the render loop
    the texels the reference's stb_image would, and — where the reference binary exists — the oracle renders the loaded
    scene to exactly the reference's PPM."""
    import json
    import shutil

    sc = sg.room_scene(300, seed=61, n_lights=3, n_materials=4, tex_size=16, n_tex_sets=1, alpha_fraction=0.0)
    path = sg.write_gltf(sc, str(tmp_path / "jpg.gltf"))
    doc = json.load(open(path))
    for i, name in enumerate(("base_420", "prog_444_full", "base_grey")):
        shutil.copy(os.path.join(JPG, name + ".jpg"), tmp_path / f"t{i}.jpg")
        doc["images"][i]["uri"] = f"t{i}.jpg"
    json.dump(doc, open(path, "w"))
    ls = rt.parse_gltf_scene(path, 64 / 48)
    tex = ls.arrays()["textures"]
    for i, name in enumerate(("base_420", "prog_444_full", "base_grey")):
        assert np.array_equal(tex[i], EXPECTED[name]), name
    if oracle.have_reference_build():
        ref = oracle.run_reference(path, 64, 48, 3, str(tmp_path / "ref.ppm"))
        fb, _ = oracle.OracleScene(ls).run_raytracer(64, 48, 3, rng_mode=rt.RT_RNG_REFERENCE)
        assert np.array_equal(oracle.tonemap(fb), ref)

"""CPU tests of the environment map (Scene::bg / Scene::bg_at, scene.h:81-89; main.cpp:28-31; config.h:36-38).

The reference switches its environment map on at compile time; tests/golden/make_envmap_golden.py let the reference's own headers
(oracle/_ref/ref_probe) do at run time what main.cpp:29-31 does under that switch and stored its answers in tests/golden/envmap/.
Here: the host HDR reader against the reference's stb_image, the oracle's bg_at and render loop against the reference's, the loader
entry point, and the exhaustive comparison of the restated atan2f / asinf (include/rt_devspec.h, what the DEVICE evaluates) with libm."""
import os
import subprocess

import numpy as np
import pytest

from conftest import golden_scene_specs, make_scene

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ENV = os.path.join(HERE, "golden", "envmap")
W, H, SPP = 64, 48, 4


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(ENV, "expected.npz"))


def with_env(sg, rt, name, picture, tmp_path):
    """A generated golden scene through the glTF loader (as the reference read it) with `picture` as its environment map."""
    sc = make_scene(sg, golden_scene_specs()[name])
    ls = rt.parse_gltf_scene(sg.write_gltf(sc, str(tmp_path / (name + ".gltf"))), W / H)
    ls.set_env_map(os.path.join(ENV, picture))
    return ls


@pytest.mark.parametrize("name", ["env_rle.hdr", "env_flat.hdr", "env_narrow.hdr", "env_wide.hdr", "env.png"])
def test_picture_decoders_match_the_reference_stb_image(rt, expected, name):
    """Radiance HDR (run-length encoded, flat with the '#?RGBE' magic, narrower than 8 pixels, 300 pixels wide with runs and dumps at their
    length limits) and the PNG: the bytes
    Texture::load_img (stbi_load, 4 channels, 8 bit: stb_image's gamma-2.2 conversion of HDR data) returned in the reference."""
    got = rt.image_decode(os.path.join(ENV, name))
    want = expected["texels_" + name.replace(".", "_")]
    assert got.shape == want.shape
    assert np.array_equal(got, want), int((got != want).sum())
    if name.endswith(".hdr"):
        assert got[..., 3].min() == 255 and got[..., :3].max() == 255  # the sun clamps
        if name != "env_wide.hdr":
            assert got[..., :3].min() == 0  # exponent 0 is black


def test_hdr_reader_refuses_what_stb_image_refuses(rt, tmp_path):
    cases = {
        "magic.hdr": b"#?RADIANT\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 1\n\x80\x80\x80\x80",
        "format.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 1 +X 1\n\x80\x80\x80\x80",
        "layout.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 1 +X 1\n\x80\x80\x80\x80",
        "layout2.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 -X 1\n\x80\x80\x80\x80",
        "scanline.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 8\n\x02\x02\x00\x09" + b"\x88\x10" * 4,
        "run.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 8\n\x02\x02\x00\x08" + b"\x89\x10" * 4,
    }
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(rt.RtError) as e:
            rt.image_decode(str(p))
        assert e.value.code == 6, name  # RT_ERR_FORMAT
    ok = tmp_path / "one.hdr"
    ok.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 1\n\x80\x40\x20\x81")  # (1, .5, .25) -> pow(v, 1/2.2) * 255 + .5
    px = rt.image_decode(str(ok))
    assert px.tolist() == [[[255, 186, 136, 255]]]


@pytest.mark.parametrize("tag,picture", [("png", "env.png"), ("hdr", "env_rle.hdr")])
def test_oracle_bg_at_matches_the_reference(rt, sg, oracle, expected, tag, picture, tmp_path):
    """Scene::bg_at for 4 027 directions (random unit vectors, the poles, the +-x / +-z axes and the atan2 / asin branch points):
    bit-identical to what the reference's own function returned."""
    orc = oracle.OracleScene(with_env(sg, rt, "open_nolight", picture, tmp_path))
    got = orc.bg_at(expected["dirs"])
    assert np.array_equal(got.view(np.uint32), expected["bg_" + tag].view(np.uint32))
    assert len(np.unique(got, axis=0)) > 1000  # really a picture, not a constant
    orc.close()
    plain = oracle.OracleScene(make_scene(sg, golden_scene_specs()["open_nolight"]))
    assert np.array_equal(plain.bg_at(expected["dirs"][:64]), np.ones((64, 3), dtype=np.float32))  # the 1x1 white default: bg_color
    plain.close()


@pytest.mark.parametrize("name,picture,ppm", [("open_nolight", "env.png", "open_nolight_envpng"), ("boxes", "env_rle.hdr", "boxes_envhdr")])
def test_oracle_render_with_environment_map_equals_the_reference_render(rt, sg, oracle, name, picture, ppm, tmp_path):
    """The reference's run_raytracer + Image::write with the map loaded (ref_probe envrender) against the oracle in reference-RNG
    mode: byte-identical PPM. The open scene ends most of its paths in the environment."""
    orc = oracle.OracleScene(with_env(sg, rt, name, picture, tmp_path))
    fb, st = orc.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_REFERENCE)
    out = tmp_path / "o.ppm"
    rt.write_ppm(str(out), rt.tonemap(fb))
    assert out.read_bytes() == open(os.path.join(ENV, f"{ppm}_{W}x{H}x{SPP}.ppm"), "rb").read()
    assert st["texel_fetches"] > 0  # the environment lookups are counted like every other Texture::sample
    orc.close()


def test_loader_attaches_an_environment_map(rt, sg, tmp_path):
    sc = make_scene(sg, golden_scene_specs()["room_textured"])
    path = sg.write_gltf(sc, str(tmp_path / "s.gltf"))
    ls = rt.parse_gltf_scene(path, W / H)
    a0 = ls.arrays()
    assert a0["bg_texture"] == -1 and np.array_equal(a0["bg_color"], [1, 1, 1])
    ls.set_env_map(os.path.join(ENV, "env_rle.hdr"), 2.5)
    a = ls.arrays()
    assert a["bg_texture"] == len(a0["textures"]) and len(a["textures"]) == len(a0["textures"]) + 1
    assert np.array_equal(a["bg_color"], np.float32([2.5, 2.5, 2.5]))
    assert np.array_equal(a["textures"][-1], rt.image_decode(os.path.join(ENV, "env_rle.hdr")))
    for t0, t1 in zip(a0["textures"], a["textures"]):
        assert np.array_equal(t0, t1)
    with pytest.raises(rt.RtError):
        ls.set_env_map(str(tmp_path / "missing.hdr"))
    txt = rt.parse_scene_txt(os.path.join(HERE, "golden", "txt", "boxes_only.txt"))
    assert txt.arrays()["bg_texture"] == -1


def test_bg_uv_restatement_equals_the_libm_expression(oracle, expected):
    """rt_bg_uv (include/rt_devspec.h: what the device evaluates, restated atan2f / asinf + the mixed float / double arithmetic of
    scene.h:85-87) against the same two lines written with libm calls, as the oracle's render loop has them: bit-identical on the golden
    directions, a million random unit vectors, non-unit vectors, and directions with |y| > 1 (NaN on both sides)."""
    rng = np.random.default_rng(11)
    d = rng.normal(size=(1_000_000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    d[:1000] *= rng.uniform(0.01, 0.9, size=(1000, 1)).astype(np.float32)  # shorter than unit length
    d[1000:1100, 1] = rng.uniform(1.0, 1.5, size=100).astype(np.float32)  # asin of > 1
    dirs = np.concatenate([expected["dirs"], d])
    a, b = oracle.bg_uv(dirs, restated=True), oracle.bg_uv(dirs, restated=False)
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert same.all(), (int((~same).sum()), dirs[(~same).any(axis=1)][:3])
    assert np.isnan(a[:, 1]).sum() >= 90 and 0.0 <= np.nanmin(a) and np.nanmax(a) <= 1.0


def test_atan2f_asinf_restatement_is_exhaustively_glibc(tmp_path):
    """rt_atanf_libm / rt_asinf_libm on all 2^32 floats, rt_atan2f_libm on 2^30 pairs + the special-operand grid, against the host's
    glibc (tools/proofs/atan2f_asinf_exhaustive.c): the device's environment lookup uses the reference's std::atan2 / std::asin."""
    exe = str(tmp_path / "atan2f_asinf_exhaustive")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "proofs", "atan2f_asinf_exhaustive.c"), "-o", exe, "-lm", "-lpthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("mismatches 0") == 4, r.stdout

"""The reference's compile-time switches (config.h:30-47) as run-time options of the host loader, each pinned to the reference: the extra light source in
camera coordinates (ADD_LIGHT_TRIANGLE, scene.h:479-498; rt_loaded_add_light_triangle) against what the reference's own constants and helpers produce and
against its render loop's output (tests/golden/light_triangle, made by ref_probe), USE_TEXTURES = false (rt_loaded_disable_textures) against the
reference's render with one-texel textures, and the CLI switches end to end. (The environment map has its own files: tests/test_envmap.py.)"""
import os

import numpy as np
import pytest

from conftest import golden_scene_specs, make_scene

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "light_triangle")
W, H, SPP = 64, 48, 4


def load_case(rt, sg, name, tmp_path):
    if name == "features":
        path = os.path.join(HERE, "golden", "features", "features.gltf")
    else:
        path = sg.write_gltf(make_scene(sg, golden_scene_specs()[name]), str(tmp_path / (name + ".gltf")))
    return rt.parse_gltf_scene(path, W / H)


@pytest.mark.parametrize("name", ["features", "open_nolight", "room_plain"])
def test_light_triangle_object_matches_the_reference_helpers(rt, sg, name, tmp_path):
    exp = np.load(os.path.join(GOLD, "expected.npz"))[name]
    ls = load_case(rt, sg, name, tmp_path)
    a0 = ls.arrays()
    ls.add_light_triangle()
    a = ls.arrays()
    n0 = a0["positions"].shape[0]
    assert a["positions"].shape[0] == n0 + 1 and len(a["materials"]) == len(a0["materials"]) + 1
    for k in ("positions", "normals", "texcoords", "tangents"):  # nothing that was there moved
        assert np.array_equal(a[k][:n0].view(np.uint32), a0[k].view(np.uint32)), k
    assert np.array_equal(a["positions"][n0].reshape(9).view(np.uint32), exp[0:9].view(np.uint32))
    assert np.array_equal(a["normals"][n0].view(np.uint32), np.tile(exp[9:12], (3, 1)).view(np.uint32))
    assert np.array_equal(a["texcoords"][n0], np.zeros((3, 2), dtype=np.float32))
    assert np.array_equal(a["tangents"][n0], np.tile(np.float32([1, 0, 0]), (3, 1)))
    m = a["materials"][a["material_ids"][n0]]
    assert np.array_equal(m["emission"], np.float32([exp[12]] * 3)) and np.array_equal(m["color"], exp[13:17])
    assert (m["roughness"], m["metallic"], m["ior"]) == (exp[17], exp[18], exp[19])
    assert m["color_tex"] == m["emissive_tex"] == m["metallic_roughness_tex"] == m["normal_tex"] == -1


@pytest.mark.parametrize("name", ["open_nolight", "features"])
def test_oracle_render_with_light_triangle_equals_the_reference_render(rt, sg, oracle, name, tmp_path):
    """The object appended as scene.h:480-497 does and rendered by the reference's run_raytracer (ref_probe lightrender) against the oracle in
    reference-RNG mode on the loader's scene + rt_loaded_add_light_triangle: byte-identical PPM (the new light also changes every light-sampling
    decision: the light BVH, uniform_int over one more light)."""
    ls = load_case(rt, sg, name, tmp_path)
    ls.add_light_triangle()
    orc = oracle.OracleScene(ls)
    fb, _ = orc.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_REFERENCE)
    out = tmp_path / "o.ppm"
    rt.write_ppm(str(out), rt.tonemap(fb))
    assert out.read_bytes() == open(os.path.join(GOLD, f"{name}_light_{W}x{H}x{SPP}.ppm"), "rb").read()
    orc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["open_nolight", "features"])
def test_device_render_with_light_triangle_equals_the_reference_bytes(gpu, sg, oracle, name, tmp_path):
    ls = load_case(gpu, sg, name, tmp_path)
    ls.add_light_triangle()
    dev = gpu.DeviceScene(ls)
    img, _ = dev.run_raytracer_rgb8(W, H, SPP, rng_mode=gpu.RT_RNG_REFERENCE)
    ref = oracle.read_ppm(os.path.join(GOLD, f"{name}_light_{W}x{H}x{SPP}.ppm"))
    assert np.array_equal(img, ref), int((img != ref).any(axis=2).sum())
    orc = oracle.OracleScene(ls)
    gfb, gst = dev.run_raytracer(W, H, 5, seed=9, counters=True)
    ofb, ost = orc.run_raytracer(W, H, 5, seed=9)
    assert np.array_equal(gfb.view(np.uint32), ofb.view(np.uint32)) and gst["light_tri_tests"] == ost["light_tri_tests"]
    dev.close()
    orc.close()


@pytest.mark.gpu
def test_cli_switches_reproduce_the_reference_renders(gpu, sg, tmp_path):
    """run.sh with RT_LIGHT_TRIANGLE=1 / RT_ENV_MAP=<picture> in reference-RNG mode writes the PPM the reference's own render loop produced with the
    corresponding compile-time switch on (ref_probe lightrender / envrender): the whole host flow (loader, switch, device render, device film, writer)."""
    import subprocess

    root = os.path.dirname(HERE)
    env = dict(os.environ, RT_RNG_MODE="reference", RT_DEVICE="0")
    out = tmp_path / "light.ppm"
    subprocess.check_call([os.path.join(root, "run.sh"), os.path.join(HERE, "golden", "features", "features.gltf"), str(W), str(H), str(SPP), str(out)],
                          env=dict(env, RT_LIGHT_TRIANGLE="1"))
    assert out.read_bytes() == open(os.path.join(GOLD, f"features_light_{W}x{H}x{SPP}.ppm"), "rb").read()
    path = sg.write_gltf(make_scene(sg, golden_scene_specs()["open_nolight"]), str(tmp_path / "open.gltf"))
    envdir = os.path.join(HERE, "golden", "envmap")
    out2 = tmp_path / "env.ppm"
    subprocess.check_call([os.path.join(root, "run.sh"), path, str(W), str(H), str(SPP), str(out2)], env=dict(env, RT_ENV_MAP=os.path.join(envdir, "env.png")))
    assert out2.read_bytes() == open(os.path.join(envdir, f"open_nolight_envpng_{W}x{H}x{SPP}.ppm"), "rb").read()
    r = subprocess.run([os.path.join(root, "run.sh"), path, str(W), str(H), "1", str(out2)], env=dict(env, RT_ENV_MAP=str(tmp_path / "missing.hdr")), capture_output=True, text=True)
    assert r.returncode == 1 and "environment map" in r.stderr


@pytest.mark.parametrize("name", ["room_textured", "features"])
def test_use_textures_false_equals_the_reference_render(rt, sg, oracle, name, tmp_path):
    """USE_TEXTURES = false (config.h:31-32): Texture::sample returns the first texel. ref_probe notexrender brings the reference's textures down to one
    texel (the code path `data.size() == 1` returns the same data[0]) and renders; the oracle on the loader's scene after rt_loaded_disable_textures
    must write the same bytes, and the image must differ from the textured one (the golden PPM of the same scene)."""
    ls = load_case(rt, sg, name, tmp_path)
    n_tex = len(ls.arrays()["textures"])
    assert n_tex >= 2
    ls.disable_textures()
    a = ls.arrays()
    assert all(t.shape == (1, 1, 4) for t in a["textures"]) and len(a["textures"]) == n_tex
    orc = oracle.OracleScene(ls)
    fb, _ = orc.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_REFERENCE)
    out = tmp_path / "o.ppm"
    rt.write_ppm(str(out), rt.tonemap(fb))
    got = out.read_bytes()
    assert got == open(os.path.join(GOLD, f"{name}_notex_{W}x{H}x{SPP}.ppm"), "rb").read()
    assert got != open(os.path.join(HERE, "golden", f"{name}_{W}x{H}x{SPP}.ppm"), "rb").read()
    orc.close()


@pytest.mark.gpu
def test_device_render_without_textures_equals_the_reference_bytes(gpu, sg, oracle, tmp_path):
    ls = load_case(gpu, sg, "room_textured", tmp_path)
    ls.disable_textures()
    dev = gpu.DeviceScene(ls)
    img, _ = dev.run_raytracer_rgb8(W, H, SPP, rng_mode=gpu.RT_RNG_REFERENCE)
    assert np.array_equal(img, oracle.read_ppm(os.path.join(GOLD, f"room_textured_notex_{W}x{H}x{SPP}.ppm")))
    dev.close()

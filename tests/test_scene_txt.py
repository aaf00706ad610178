"""The scene-txt front end (csrc/host/txt_loader.cpp; BASELINE configs 1-2). CPU only.

The reference at HEAD cannot read these files (no parser, triangles only), so:
  * TRIANGLE / BOX primitives become triangles and ARE pinned to the reference: the same arrays exported as glTF load
    back bit-identically through the glTF loader and render to the golden PPM the unmodified reference produced
    (tests/golden/txt_boxes_64x48x4.ppm, made by make_golden.py); where oracle/_ref exists the reference runs live too;
  * ELLIPSOID / PLANE are analytic primitives whose semantics are this project's (include/rt_primspec.h): "parity
    unpinned" — checked for self-consistency here and for oracle == GPU in tests/test_gpu_txt.py.
"""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TXT = os.path.join(GOLD, "txt")
W, H, SPP = 64, 48, 4


def test_grammar_and_material_mapping(rt):
    ls = rt.parse_scene_txt(os.path.join(TXT, "cornell_mixed.txt"))
    a = ls.arrays()
    assert ls.info() == {"width": 96, "height": 96, "samples": 16, "ignored_lights": 1}
    assert a["ray_depth"] == 6 and np.allclose(a["bg_color"], [0.05, 0.05, 0.1])
    assert a["positions"].shape[0] == 12 + 12 + 1 and len(a["primitives"]) == 8  # 2 boxes + 1 triangle; 5 planes + 3 ellipsoids
    assert [p["kind"] for p in a["primitives"]] == [2, 2, 2, 2, 2, 1, 1, 1]
    # one material per primitive, in file order; METALLIC -> metallic 1 / roughness 0; DIELECTRIC -> roughness 0 + ior
    m = a["materials"]
    assert len(m) == 11
    assert np.allclose(m[3]["color"], [1, 0.25, 0.25, 1]) and m[3]["metallic"] == 0 and m[3]["roughness"] == 1
    assert np.allclose(m[5]["emission"], [3, 3, 3]) and np.allclose(m[5]["color"], [0, 0, 0, 1])  # COLOR defaults to black
    assert m[7]["metallic"] == 1 and m[7]["roughness"] == 0
    assert m[8]["metallic"] == 0 and m[8]["roughness"] == 0 and np.isclose(m[8]["ior"], 1.33)
    # a plane's ROTATION turns its normal once, at load: "PLANE 0 0 3" rotated -90 degrees about y -> (-3, 0, 0)
    p = a["primitives"][4]
    assert np.allclose(p["param"], [-3, 0, 0], atol=1e-5) and np.allclose(p["rotation"], [0, 0, 0, 1])
    # BOX: 12 triangles, outward winding (face normal points away from the box centre), rotated + translated
    box = a["positions"][12:24]
    centre = np.array([-2, -2.2, -1], dtype=np.float32)
    n = np.cross(box[:, 1] - box[:, 0], box[:, 2] - box[:, 0])
    assert (np.einsum("ij,ij->i", n, box.mean(axis=1) - centre) > 0).all()
    assert np.allclose(np.linalg.norm(a["normals"], axis=2), 1, atol=1e-6)
    assert (a["tangents"] == np.array([1, 0, 0], dtype=np.float32)).all() and (a["texcoords"] == 0).all()
    cam = a["camera"]
    assert np.allclose(cam["position"], [0, 0, 14]) and np.isclose(cam["fov_x"], 0.927295218)
    # rt_scene_load dispatches on the extension (what the CLI calls)
    assert rt.load_scene(os.path.join(TXT, "cornell_mixed.txt"), 1.0).arrays()["positions"].shape == a["positions"].shape


@pytest.mark.parametrize("text,needle", [
    ("NEW_PRIMITIVE\nSPHERE 1 1 1\n", "unknown command 'SPHERE'"),
    ("POSITION 0 0 0\n", "outside NEW_PRIMITIVE"),
    ("NEW_PRIMITIVE\nBOX 1 x 1\n", "not a number"),
    ("NEW_PRIMITIVE\nELLIPSOID 1 1\n", "expects 3 numbers"),
    ("NEW_PRIMITIVE\nCOLOR 1 1 1\nNEW_PRIMITIVE\nBOX 1 1 1\n", "without ELLIPSOID"),
    ("RAY_DEPTH 99\n", "RAY_DEPTH out of range"),
    ("RAY_DEPTH 2.7\n", "RAY_DEPTH out of range"),
    ("SAMPLES -4\n", "SAMPLES out of range"),
    ("SAMPLES nan\n", "SAMPLES out of range"),
    ("DIMENSIONS 640 1e30\n", "DIMENSIONS out of range"),
    ("\n\n", "empty scene file"),
])
def test_errors_are_codes_with_line_numbers(rt, tmp_path, text, needle):
    f = tmp_path / "bad.txt"
    f.write_text(text)
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene_txt(str(f))
    assert e.value.code == 6 and needle in str(e.value), str(e.value)  # RT_ERR_FORMAT
    with pytest.raises(rt.RtError) as e:
        rt.parse_scene_txt(str(tmp_path / "missing.txt"))
    assert e.value.code == 5  # RT_ERR_IO


def _txt_boxes(rt, sg, tmp_path):
    ls = rt.parse_scene_txt(os.path.join(TXT, "boxes_only.txt"))
    a = ls.arrays()
    sc = sg.scene_from_arrays(a, yfov=0.8, rotation=(0.0, 0.0, 0.0, 1.0), face_normals=True)
    path = sg.write_gltf(sc, str(tmp_path / "txt_boxes.gltf"))
    return ls, a, path


def test_box_and_triangle_primitives_are_pinned_through_gltf(rt, sg, oracle, tmp_path):
    """boxes_only.txt (BOX + TRIANGLE only): its triangles exported as glTF come back bit-identical through the glTF loader
    (so both front ends feed the render loop the same scene), and the oracle renders the txt-loaded scene to exactly the
    PPM the unmodified reference binary produced from that glTF (golden fixture)."""
    ls, a, path = _txt_boxes(rt, sg, tmp_path)
    assert a["positions"].shape[0] == 5 * 12 + 2 and len(a["primitives"]) == 0
    g = rt.parse_gltf_scene(path, W / H).arrays()
    for k in ("positions", "normals", "texcoords", "tangents"):
        assert np.array_equal(a[k].view(np.uint32), g[k].view(np.uint32)), k
    assert np.array_equal(a["material_ids"], g["material_ids"])
    for ma, mg in zip(a["materials"], g["materials"]):
        for k in ("color", "emission", "roughness", "metallic", "ior"):
            assert np.array_equal(np.asarray(ma[k]), np.asarray(mg[k])), k
    for k in ("position", "right", "up", "forward", "fov_x"):
        assert np.array_equal(np.asarray(a["camera"][k], dtype=np.float32).view(np.uint32), np.asarray(g["camera"][k], dtype=np.float32).view(np.uint32)), k
    orc = oracle.OracleScene(ls)
    fb, _ = orc.run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_REFERENCE)
    out = tmp_path / "o.ppm"
    rt.write_ppm(str(out), rt.tonemap(fb))
    assert out.read_bytes() == open(os.path.join(GOLD, f"txt_boxes_{W}x{H}x{SPP}.ppm"), "rb").read()
    if oracle.have_reference_build():  # and live, where the reference binary exists (this container)
        ref = oracle.run_reference(path, W, H, SPP, str(tmp_path / "ref.ppm"))
        assert np.array_equal(oracle.tonemap(fb), ref)


def test_config1_fixture_and_its_box_pinned_to_the_reference_binary(rt, sg, oracle, tmp_path):
    """BASELINE config 1 = sample_data/scene-000.txt (ELLIPSOID, PLANE, BOX; 256x256, 4 SPP). The committed fixture is that file; the part
    the reference at HEAD can still render — the BOX's 12 triangles — comes back bit-identical through the glTF loader and the oracle renders
    it to the bytes the UNMODIFIED reference binary produced (tests/golden/make_scene000_golden.py). The GPU half: tests/test_gpu_txt.py."""
    from conftest import SCENE000, scene000_box_gltf

    ref_file = "/root/reference/sample_data/scene-000.txt"
    if os.path.exists(ref_file):
        assert open(ref_file, "rb").read() == open(SCENE000, "rb").read()
    gltf, a = scene000_box_gltf(rt, sg, tmp_path)
    assert a["positions"].shape[0] == 12 and [p["kind"] for p in a["primitives"]] == [1, 2]  # BOX -> 12 triangles; ELLIPSOID, PLANE
    ls = rt.parse_gltf_scene(gltf, W / H)
    g = ls.arrays()
    for k in ("positions", "normals", "texcoords", "tangents"):
        assert np.array_equal(a[k].view(np.uint32), g[k].view(np.uint32)), k
    for k in ("color", "emission", "roughness", "metallic", "ior"):
        assert np.array_equal(np.asarray(a["materials"][int(a["material_ids"][0])][k]), np.asarray(g["materials"][int(g["material_ids"][0])][k])), k
    fb, _ = oracle.OracleScene(ls).run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_REFERENCE)
    out = tmp_path / "o.ppm"
    rt.write_ppm(str(out), rt.tonemap(fb))
    assert out.read_bytes() == open(os.path.join(GOLD, f"txt_scene000_box_{W}x{H}x{SPP}.ppm"), "rb").read()
    if oracle.have_reference_build():
        assert np.array_equal(oracle.tonemap(fb), oracle.run_reference(gltf, W, H, SPP, str(tmp_path / "ref.ppm")))
    # the whole file at the configuration's size on the CPU path (reference RNG, as config 1 says): finite, deterministic, all three kinds visible
    orc = oracle.OracleScene(rt.parse_scene_txt(SCENE000))
    fb1, st = orc.run_raytracer(256, 256, 4, rng_mode=rt.RT_RNG_REFERENCE, threads=1)
    fb2, _ = orc.run_raytracer(256, 256, 4, rng_mode=rt.RT_RNG_REFERENCE, threads=4)
    assert st["samples"] == 256 * 256 * 4 and np.isfinite(fb1).all() and np.array_equal(fb1.view(np.uint32), fb2.view(np.uint32))
    prim, _ = orc.cast_rays(_primary_rays(a["camera"], 256, 256))
    assert set(np.unique(prim).tolist()) >= {12, 13} and (prim < 12).sum() > 100  # ELLIPSOID (12), PLANE (13) and BOX triangles all seen
    orc.close()


def _primary_rays(cam, W_, H_):
    """Pixel-centre rays of gen_ray (raytracer.h:516-525) for a loaded camera, float32."""
    x, y = np.meshgrid(np.arange(W_, dtype=np.float64) + 0.5, np.arange(H_, dtype=np.float64) + 0.5)
    tx = np.tan(float(cam["fov_x"]) / 2)
    ty = tx * H_ / W_
    d = ((2 * x / W_ - 1) * tx)[..., None] * cam["right"].astype(np.float64) - ((2 * y / H_ - 1) * ty)[..., None] * cam["up"].astype(np.float64) + cam["forward"].astype(np.float64)
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    o = np.broadcast_to(cam["position"].astype(np.float64), d.shape)
    return np.concatenate([o, d], axis=2).reshape(-1, 6).astype(np.float32)


def test_analytic_primitives_self_consistency(rt, oracle):
    """ELLIPSOID / PLANE ("parity unpinned"): closest hits reported through the oracle's probe agree with an independent
    double-precision evaluation of the same geometry, and the front / back conventions hold."""
    ls = rt.parse_scene_txt(os.path.join(TXT, "cornell_mixed.txt"))
    a = ls.arrays()
    orc = oracle.OracleScene(ls)
    rng = np.random.default_rng(5)
    n = 4000
    o = rng.uniform(-4, 4, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    prim, bct = orc.cast_rays(rays)
    n_tri = a["positions"].shape[0]
    assert (prim != 0xFFFFFFFF).all()  # a closed room of planes: every ray hits something
    hit_prims = prim[prim >= n_tri] - n_tri
    assert len(np.unique(hit_prims)) == 8  # every analytic primitive is hit by some ray

    def quat_rot(q, v):
        u, w = np.asarray(q[:3], dtype=np.float64), float(q[3])
        return v + 2 * w * np.cross(u, v) + 2 * np.cross(u, np.cross(u, v))

    r64 = rays.astype(np.float64)
    checked = 0
    for i in range(n):
        if prim[i] < n_tri:
            continue
        p = a["primitives"][prim[i] - n_tri]
        oo, dd, t = r64[i, :3], r64[i, 3:], float(bct[i, 2])
        x = oo + dd * t
        if p["kind"] == 2:
            nn = p["param"].astype(np.float64)
            assert abs(np.dot(x - p["position"], nn / np.linalg.norm(nn))) < 1e-3
        else:
            q = p["rotation"].astype(np.float64)
            loc = quat_rot(np.array([-q[0], -q[1], -q[2], q[3]]), x - p["position"])
            assert abs(np.sum((loc / p["param"]) ** 2) - 1) < 2e-3
        assert t >= 1e-4
        checked += 1
    assert checked > 1000
    # a render of the mixed scene is finite and deterministic, independent of the thread count
    f1, s1 = orc.run_raytracer(48, 48, 4, seed=3, threads=1)
    f2, s2 = orc.run_raytracer(48, 48, 4, seed=3, threads=5)
    assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32)) and np.isfinite(f1).all() and s1["casts"] == s2["casts"]


def test_config2_fixture_is_the_reference_file_and_renders(rt, oracle):
    """BASELINE config 2 names sample_data/homebrew_primitives (spheres / boxes, 512x512, 64 SPP). The committed fixture is one of those files,
    practice3_5.txt, byte for byte (checked where /root/reference exists); it parses to 24 BOX triangles + 6 analytic primitives with the file's
    own size, depth and sample count, and the CPU oracle renders it (a closed, lit Cornell box: finite, not black)."""
    from conftest import PRACTICE3_5

    ref_file = "/root/reference/sample_data/homebrew_primitives/practice3_5.txt"
    if os.path.exists(ref_file):
        assert open(ref_file, "rb").read() == open(PRACTICE3_5, "rb").read()
    ls = rt.parse_scene_txt(PRACTICE3_5)
    a = ls.arrays()
    assert a["positions"].shape[0] == 24 and [p["kind"] for p in a["primitives"]] == [2, 2, 2, 2, 2, 1]
    info = ls.info()
    assert (info["width"], info["height"], info["samples"], int(a["ray_depth"])) == (512, 512, 64, 6)
    fb, st = oracle.OracleScene(ls).run_raytracer(128, 128, 8, rng_mode=rt.RT_RNG_REFERENCE)
    assert st["samples"] == 128 * 128 * 8 and np.isfinite(fb).all() and 0.01 < fb.mean() < 10


def test_reference_sample_data_parses_if_present(rt, oracle):
    """Every scene file the reference ships under sample_data/ (BASELINE config 1 names scene-000.txt) goes through the
    loader and renders on the CPU oracle. Container only: /root/reference does not exist on the GPU box."""
    files = sorted(glob.glob("/root/reference/sample_data/*.txt") + glob.glob("/root/reference/sample_data/homebrew_primitives/*.txt"))
    if not files:
        pytest.skip("/root/reference/sample_data absent")
    assert len(files) == 13
    for f in files:
        ls = rt.parse_scene_txt(f)
        a = ls.arrays()
        assert a["positions"].shape[0] + len(a["primitives"]) > 0
        if f.endswith("scene-000.txt"):  # BASELINE config 1: 256x256, 4 SPP on the CPU path
            assert a["positions"].shape[0] == 12 and [p["kind"] for p in a["primitives"]] == [1, 2]
            fb, st = oracle.OracleScene(ls).run_raytracer(256, 256, 4, rng_mode=rt.RT_RNG_REFERENCE)
            assert st["samples"] == 256 * 256 * 4 and np.isfinite(fb).all() and 0 < fb.mean() < 1


def _sphere_scene(sg, radius):
    """No triangles, one ELLIPSOID with equal semi-axes at the origin, identity rotation."""
    mat = sg.Material(color=(0.8, 0.8, 0.8, 1.0), emission=(0.0, 0.0, 0.0))
    z = np.zeros((0, 3, 3), dtype=np.float32)
    return sg.Scene(positions=z, normals=z, texcoords=np.zeros((0, 3, 2), dtype=np.float32), tangents=z, material_ids=np.zeros(0, dtype=np.uint32), materials=[mat],
                    camera=sg.look_camera((0.0, 0.0, 5.0), yaw_deg=0.0, yfov=0.8),
                    primitives=[dict(kind=1, material_id=0, param=(radius,) * 3, position=(0.0, 0.0, 0.0), rotation=(0.0, 0.0, 0.0, 1.0))])


def sphere_kat_check(cast, sg, kat, k):
    """cast(scene, rays) -> (prim, bct). The primitive's distance is the reference function's first root >= EPS (raytracer.h:61-77), bit for bit."""
    EPS = np.float32(1e-4)
    rays, radius, t12 = kat[f"rays_{k}"], float(kat[f"radius_{k}"]), kat[f"t_{k}"]
    prim, bct = cast(_sphere_scene(sg, radius), rays)
    t1, t2 = t12[:, 0], t12[:, 1]
    use1 = t1 >= EPS
    use2 = ~use1 & (t2 >= EPS)
    want_hit = use1 | use2
    want_t = np.where(use1, t1, t2)
    assert np.array_equal(prim != 0xFFFFFFFF, want_hit), int(((prim != 0xFFFFFFFF) != want_hit).sum())
    got_t = bct[:, 2]
    assert np.array_equal(got_t[want_hit].view(np.uint32), want_t[want_hit].view(np.uint32))
    assert want_hit.sum() > 1000 and use2.sum() > 300 and (~want_hit).sum() > 1000  # outside hits, inside hits, misses all present


@pytest.mark.parametrize("k", [0, 1, 2])
def test_ellipsoid_solve_is_the_reference_sphere_routine(rt, sg, oracle, k):
    """The ELLIPSOID's quadratic solve restates the reference's (unused) intersect_ray_sphere, raytracer.h:61-77: for a sphere at the origin
    its distance is that function's root, bit for bit (tests/golden/sphere_kat.npz, computed BY the reference function through ref_probe).
    This pins the solver; translation, rotation, root choice and normal remain definitions of include/rt_primspec.h."""
    kat = np.load(os.path.join(GOLD, "sphere_kat.npz"))

    def cast(scene, rays):
        orc = oracle.OracleScene(scene)
        try:
            return orc.cast_rays(rays)
        finally:
            orc.close()

    sphere_kat_check(cast, sg, kat, k)

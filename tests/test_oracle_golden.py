"""Pins the CPU oracle (oracle/rt_oracle.cpp) and the host loader against golden vectors manufactured from the
UNMODIFIED reference (tests/golden/make_golden.py). CPU only.

Parity chain, hop 1: oracle in reference-RNG mode + libm sin/cos == the reference binary, byte for
byte in the PPM, bit for bit in BVH nodes, closest hits (prim, b, c, t) and light pdf values.
"""
import os

import numpy as np
import pytest

from conftest import golden_scene_specs, make_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GENERATED = list(golden_scene_specs().keys())
# "features": hand-built glTF committed under tests/golden/features/ (scene selection, matrix + TRS nodes, strips,
# u8/u16/u32 indices, missing attributes, lower-case `tangent`, emissive texture + strength, ...; make_features_gltf.py)
NAMES = GENERATED + ["features"]
W, H, SPP = 64, 48, 4


@pytest.fixture(scope="module")
def loaded(rt, sg, tmp_path_factory):
    """Each golden scene written as glTF and read back through the C++ host loader (no GPU involved)."""
    out = {}
    d = tmp_path_factory.mktemp("gltf")
    for name, spec in golden_scene_specs().items():
        sc = make_scene(sg, spec)
        path = sg.write_gltf(sc, str(d / (name + ".gltf")))
        out[name] = (rt.parse_gltf_scene(path, W / H), sc, path)
    path = os.path.join(GOLD, "features", "features.gltf")
    out["features"] = (rt.parse_gltf_scene(path, W / H), None, path)
    return out


@pytest.fixture(scope="module")
def oracles(oracle, loaded):
    return {name: oracle.OracleScene(ls) for name, (ls, _, _) in loaded.items()}


def gold(name):
    return np.load(os.path.join(GOLD, f"{name}_probe.npz"))


@pytest.mark.parametrize("name", NAMES)
def test_loader_matches_reference_scene_objects(loaded, name):
    """C++ glTF loader == parse_gltf_scene (scene.h:183): every triangle attribute and the camera, bit for bit."""
    ls, _, _ = loaded[name]
    a = ls.arrays()
    g = gold(name)
    n = a["positions"].shape[0]
    assert n == g["obj_positions"].shape[0]
    assert np.array_equal(a["positions"].reshape(n, 9).view(np.uint32), g["obj_positions"].view(np.uint32))
    assert np.array_equal(a["normals"].reshape(n, 9).view(np.uint32), g["obj_normals"].view(np.uint32))
    assert np.array_equal(a["texcoords"].reshape(n, 6).view(np.uint32), g["obj_texcoords"].view(np.uint32))
    assert np.array_equal(a["tangents"].reshape(n, 9).view(np.uint32), g["obj_tangents"].view(np.uint32))
    cam = np.concatenate([a["camera"]["position"], a["camera"]["right"], a["camera"]["up"], a["camera"]["forward"], [a["camera"]["fov_x"]]]).astype(np.float32)
    assert np.array_equal(cam.view(np.uint32), g["camera"].view(np.uint32))


@pytest.mark.parametrize("name", GENERATED)
def test_loader_matches_python_generator(loaded, rt, name):
    """The direct-ABI path (numpy arrays -> rt_scene_desc, used by bench.py) feeds the same triangles as the glTF path."""
    ls, sc, _ = loaded[name]
    a = ls.arrays()
    n = sc.n_triangles
    assert np.array_equal(a["positions"].view(np.uint32), sc.positions.astype(np.float32).view(np.uint32))
    assert np.array_equal(a["normals"].view(np.uint32), sc.resolved_normals().view(np.uint32))
    assert np.array_equal(a["texcoords"].view(np.uint32), sc.texcoords.astype(np.float32).view(np.uint32))
    used = sorted(set(int(m) for m in sc.material_ids))
    # glTF materials are converted lazily in order of first use; compare through the per-triangle indirection
    for t in range(0, n, max(1, n // 50)):
        gm = a["materials"][int(a["material_ids"][t])]
        pm = sc.materials[int(sc.material_ids[t])]
        assert np.array_equal(gm["color"], np.asarray(pm.color, dtype=np.float32))
        assert np.array_equal(gm["emission"], pm.emission_f32())
        assert gm["roughness"] == np.float32(pm.roughness) and gm["metallic"] == np.float32(pm.metallic)
        assert (gm["color_tex"], gm["normal_tex"], gm["metallic_roughness_tex"], gm["emissive_tex"]) == (pm.color_tex, pm.normal_tex, pm.metallic_roughness_tex, pm.emissive_tex)
    assert len(used) >= 2
    for i, t in enumerate(sc.textures):
        assert np.array_equal(a["textures"][i], t)
    for k in ("position", "right", "up", "forward"):
        assert np.array_equal(a["camera"][k], np.asarray(getattr(sc.camera, k), dtype=np.float32)), k


@pytest.mark.parametrize("name", NAMES)
def test_oracle_bvh_matches_reference(oracles, name):
    """BVH::build (bvh.h:368-393): node boxes, links, leaf ranges and the object permutation, for both BVHs."""
    g = gold(name)
    for which, tag in ((0, "scene"), (1, "light")):
        b = oracles[name].bvh_info(which)
        assert b["root"] == int(g[f"{tag}_root"])
        assert np.array_equal(b["order"], g[f"{tag}_order"])
        assert np.array_equal(b["nodes"], g[f"{tag}_nodes"])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_closest_hits_match_reference(oracles, name):
    """BVH::intersect_ray (bvh.h:195-235): bit-exact (prim, b, c, t) for pixel-centre rays and random rays."""
    g = gold(name)
    for rays, prim, bct in ((g["primary_rays"], g["primary_prim"], g["primary_bct"]), (g["cast_rays"], g["cast_prim"], g["cast_bct"])):
        p, b = oracles[name].cast_rays(rays)
        assert np.array_equal(p, prim)
        assert np.array_equal(b.view(np.uint32), bct.view(np.uint32))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_light_pdf_matches_reference(oracles, name):
    """bvh_mix_dist::pdf (raytracer.h:363-375) through foreach_intersection (bvh.h:237-260)."""
    g = gold(name)
    got = oracles[name].light_pdf(g["cast_rays"])
    assert np.array_equal(got.view(np.uint32), g["light_pdf"].view(np.uint32))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_ppm_byte_identical_to_reference(oracles, oracle, rt, name, tmp_path):
    """run_raytracer + Image::set_pixel + Image::write: the PPM the reference binary produced, byte for byte."""
    fb, st = oracles[name].run_raytracer(W, H, SPP, rng_mode=rt.RT_RNG_REFERENCE)
    img = oracle.tonemap(fb)
    ref = oracle.read_ppm(os.path.join(GOLD, f"{name}_{W}x{H}x{SPP}.ppm"))
    assert img.shape == ref.shape
    assert np.array_equal(img, ref), f"{int((img != ref).any(axis=2).sum())} pixels differ"
    # the product's host film + PPM writer produce the same file
    out = tmp_path / "o.ppm"
    rt.write_ppm(str(out), rt.tonemap(fb))
    assert out.read_bytes() == open(os.path.join(GOLD, f"{name}_{W}x{H}x{SPP}.ppm"), "rb").read()
    assert st["samples"] == W * H * SPP


@pytest.mark.parametrize("name", NAMES)
def test_product_host_bvh_builder_matches_reference(rt, loaded, name):
    """The product's own builder (csrc/bvh_build.cpp, what rt_create uploads) against the reference's BVH dump."""
    ls, sc, _ = loaded[name]
    g = gold(name)
    a = ls.arrays()
    b = rt.bvh_build_host(a["positions"])
    assert b["root"] == int(g["scene_root"])
    assert np.array_equal(b["nodes"], g["scene_nodes"]) and np.array_equal(b["order"], g["scene_order"])
    lights = np.array([i for i, m in enumerate(a["material_ids"]) if np.any(a["materials"][int(m)]["emission"] != 0)], dtype=np.uint32)
    bl = rt.bvh_build_host(a["positions"], lights)
    assert np.array_equal(bl["nodes"], g["light_nodes"]) and np.array_equal(bl["order"], g["light_order"])


def test_parallel_subtree_build_is_bit_identical(rt, sg, oracle, monkeypatch):
    """Above 65 536 triangles the product's builder AND the oracle split the top of the tree into parallel tasks; node
    numbering, boxes and the object permutation must equal the sequential, literal oracle build (plain build_node)."""
    sc = sg.room_scene(150000, seed=3, n_lights=8, tex_size=0)
    b = rt.bvh_build_host(sc.positions)
    monkeypatch.setenv("RTO_BUILD_PARALLEL_LEVELS", "0")
    ob = oracle.OracleScene(sc).bvh_info(0)
    monkeypatch.setenv("RTO_BUILD_PARALLEL_LEVELS", "4")
    op = oracle.OracleScene(sc).bvh_info(0)
    assert b["root"] == ob["root"] == op["root"]
    assert np.array_equal(b["nodes"], ob["nodes"]) and np.array_equal(b["order"], ob["order"])
    assert np.array_equal(op["nodes"], ob["nodes"]) and np.array_equal(op["order"], ob["order"])
    leaves = b["nodes"][b["nodes"][:, 6] == 0xFFFFFFFF]
    assert (leaves[:, 9] - leaves[:, 8]).sum() == sc.n_triangles


def test_live_reference_bvh_dump_at_100k_if_present(oracle, rt, sg, tmp_path):
    """The reference's own BVH::build (through oracle/_ref/ref_probe) on 100 000 triangles against the oracle's parallel
    build and the product's host builder: nodes and object order bit for bit. Container only (needs oracle/_ref)."""
    if not oracle.have_reference_build():
        pytest.skip("oracle/_ref not built (GPU box): covered by the committed fixtures")
    sc = sg.room_scene(100000, seed=99, n_lights=20, n_materials=4, tex_size=0, offset=0.05)
    path = sg.write_gltf(sc, str(tmp_path / "big.gltf"))
    oracle.ref_probe("bvh", path, 64, 48, str(tmp_path / "bvh.bin"))
    words = np.fromfile(str(tmp_path / "bvh.bin"), dtype=np.uint32)
    ls = rt.parse_gltf_scene(path, 64 / 48)
    orc = oracle.OracleScene(ls)
    a = ls.arrays()
    lights = np.array([i for i, m in enumerate(a["material_ids"]) if np.any(a["materials"][int(m)]["emission"] != 0)], dtype=np.uint32)
    p = 0
    for which, sub in ((0, None), (1, lights)):
        nn, no, root = (int(x) for x in words[p : p + 3])
        p += 3
        nodes = words[p : p + 10 * nn].reshape(nn, 10)
        p += 10 * nn
        order = words[p : p + no]
        p += no
        ob = orc.bvh_info(which)
        pb = rt.bvh_build_host(a["positions"], sub)
        assert ob["root"] == root == pb["root"]
        assert np.array_equal(ob["nodes"], nodes) and np.array_equal(ob["order"], order), which
        assert np.array_equal(pb["nodes"], nodes) and np.array_equal(pb["order"], order), which
    assert nn > 1  # 20 lights: a light BVH with inner nodes


def test_oracle_thread_count_does_not_change_the_image(oracles, rt):
    a, _ = oracles["room_plain"].run_raytracer(W, H, 2, rng_mode=rt.RT_RNG_REFERENCE, threads=1)
    b, _ = oracles["room_plain"].run_raytracer(W, H, 2, rng_mode=rt.RT_RNG_REFERENCE, threads=7)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_oracle_sharding_union(oracles, rt):
    full, _ = oracles["boxes"].run_raytracer(W, H, 2, seed=5)
    fb = np.full((H, W, 3), -1, dtype=np.float32)
    for r in range(3):
        oracles["boxes"].run_raytracer(W, H, 2, seed=5, shard_index=r, shard_count=3, shard_block=256, out=fb)
    assert np.array_equal(fb.view(np.uint32), full.view(np.uint32))


def test_live_reference_binary_if_present(oracle, rt, sg, tmp_path):
    """Where oracle/_ref exists (this container), run the reference binary on a fresh scene that is NOT a fixture."""
    if not oracle.have_reference_build():
        pytest.skip("oracle/_ref not built (GPU box): covered by the committed fixtures")
    sc = sg.room_scene(900, seed=777, n_lights=5, n_materials=7, tex_size=8, n_tex_sets=3, alpha_fraction=0.3, smooth_normals=True)
    path = sg.write_gltf(sc, str(tmp_path / "live.gltf"))
    ref = oracle.run_reference(path, 40, 56, 3, str(tmp_path / "ref.ppm"))
    ls = rt.parse_gltf_scene(path, 40 / 56)
    orc = oracle.OracleScene(ls)
    fb, _ = orc.run_raytracer(40, 56, 3, rng_mode=rt.RT_RNG_REFERENCE)
    assert np.array_equal(oracle.tonemap(fb), ref)


def test_live_reference_binary_on_a_deep_tree_if_present(oracle, rt, sg, tmp_path):
    """Same as above at 30 000 triangles (a BVH ~20 levels deep, textured, dense overlaps): the oracle's PPM must still be
    byte-identical to the reference binary's. Container only (needs oracle/_ref)."""
    if not oracle.have_reference_build():
        pytest.skip("oracle/_ref not built (GPU box): covered by the committed fixtures")
    sc = sg.room_scene(30000, seed=4711, n_lights=6, n_materials=12, tex_size=16, n_tex_sets=4, alpha_fraction=0.1, offset=0.4)
    path = sg.write_gltf(sc, str(tmp_path / "deep.gltf"))
    ref = oracle.run_reference(path, 96, 64, 2, str(tmp_path / "ref.ppm"))
    ls = rt.parse_gltf_scene(path, 96 / 64)
    fb, _ = oracle.OracleScene(ls).run_raytracer(96, 64, 2, rng_mode=rt.RT_RNG_REFERENCE)
    assert np.array_equal(oracle.tonemap(fb), ref)

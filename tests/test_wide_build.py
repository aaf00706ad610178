"""CPU tests of the production build (RT_BUILD_WIDE): the 8-wide tree with quantised child boxes that wide_build.cpp
collapses out of the reference-topology binary tree (src/bvh.h:262-393 gives that one). No GPU: rt_bvh_wide_build_host.

The tree is not the reference's, so it cannot be compared with it node by node; what must hold for the traversal
(csrc/rt_wide.hip) to find every hit the reference finds is checked instead, in exact (float64) arithmetic:
every triangle sits in exactly one leaf slot, every slot's quantised box contains everything below it, inner children and
leaf triangles are laid out where the kernel's index arithmetic looks for them, empty slots cannot be hit."""
import numpy as np
import pytest

from conftest import golden_scene_specs, make_scene


def decode(nodes):
    """(n, 20) u32 WideNode records -> dict of arrays."""
    b = nodes.view(np.uint8).reshape(len(nodes), 80)
    return dict(
        p=nodes[:, 0:3].copy().view(np.float32),
        e=b[:, 12:15].astype(np.int32),
        imask=b[:, 15].astype(np.uint32),
        child_base=nodes[:, 4],
        tri_base=nodes[:, 5],
        tri_mask=nodes[:, 6],
        qlo=b[:, 32:56].reshape(-1, 3, 8).astype(np.float64),
        qhi=b[:, 56:80].reshape(-1, 3, 8).astype(np.float64),
    )


def walk_and_check(nodes, order, positions):
    """Returns (depth, triangles per leaf slot histogram). Raises AssertionError on any structural violation."""
    n_nodes, n_tris = len(nodes), len(order)
    if n_tris == 0:
        assert n_nodes == 0
        return 0, {}
    D = decode(nodes)
    pos = positions.reshape(-1, 3, 3).astype(np.float64)
    tri_lo, tri_hi = pos.min(axis=1), pos.max(axis=1)
    seen_tri = np.zeros(n_tris, dtype=np.int32)
    seen_node = np.zeros(n_nodes, dtype=np.int32)
    seen_node[0] = 1
    hist = {}

    def visit(i, depth):
        """returns (lo, hi) of everything below node i, exact"""
        cell = np.ldexp(1.0, D["e"][i] - 127)  # per axis
        p = D["p"][i].astype(np.float64)
        imask, tmask = int(D["imask"][i]), int(D["tri_mask"][i])
        assert tmask >> 24 == 0
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        max_depth = depth
        rank_inner = 0
        for s in range(8):
            qlo, qhi = D["qlo"][i, :, s], D["qhi"][i, :, s]
            blo, bhi = p + qlo * cell, p + qhi * cell
            tbits = (tmask >> (3 * s)) & 7
            if imask >> s & 1:
                assert tbits == 0, "a slot is inner or leaf, not both"
                c = int(D["child_base"][i]) + rank_inner
                rank_inner += 1
                assert 0 < c < n_nodes and seen_node[c] == 0, "inner children: consecutive, each node referenced once"
                seen_node[c] = 1
                clo, chi, d = visit(c, depth + 1)
                max_depth = max(max_depth, d)
            elif tbits:
                assert tbits in (1, 3, 7), "a leaf slot's triangles are its first bits"
                cnt = bin(tbits).count("1")
                hist[cnt] = hist.get(cnt, 0) + 1
                first = int(D["tri_base"][i]) + bin(tmask & ((1 << (3 * s)) - 1)).count("1")
                ks = np.arange(first, first + cnt)
                assert ks.max() < n_tris
                seen_tri[ks] += 1
                t = order[ks]
                clo, chi = tri_lo[t].min(axis=0), tri_hi[t].max(axis=0)
            else:
                assert (qlo > qhi).all() or (qlo == 255).all(), "an empty slot carries an inverted box"
                assert (qlo == 255).all() and (qhi == 0).all()
                continue
            assert (blo <= clo).all() and (bhi >= chi).all(), f"node {i} slot {s}: the quantised box does not contain its contents"
            # ... and is not sloppy: within one grid cell of the exact box
            assert (clo - blo < cell * 1.0000001).all() and (bhi - chi < cell * 1.0000001).all()
            lo_all, hi_all = np.minimum(lo_all, clo), np.maximum(hi_all, chi)
        # the grid origin is the node's own lower corner and 255 cells span the node
        assert (p <= lo_all).all() and (p + 255 * cell >= hi_all).all()
        return lo_all, hi_all, max_depth

    import sys

    sys.setrecursionlimit(10000)
    _, _, depth = visit(0, 1)
    assert (seen_tri == 1).all(), f"{int((seen_tri != 1).sum())} triangle records are not in exactly one leaf slot"
    assert sorted(order.tolist()) == list(range(n_tris)), "order is a permutation of the triangles"
    assert seen_node.all(), "unreachable wide nodes"
    return depth, hist


@pytest.mark.parametrize("name", ["room_plain", "boxes", "open_nolight"])
def test_wide_tree_is_well_formed(rt, sg, name):
    sc = make_scene(sg, golden_scene_specs()[name])
    w = rt.bvh_wide_build_host(sc.positions)
    depth, hist = walk_and_check(w["nodes"], w["order"], sc.positions)
    assert depth == w["depth"]
    assert set(hist) <= {1, 2, 3}
    n_tris = len(w["order"])
    assert len(w["nodes"]) < n_tris  # really wide: far fewer nodes than a binary tree's n - 1
    assert w["sah_cost"] > 0


def test_wide_tree_tiny_and_degenerate_scenes(rt):
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 4, 9, 17):
        pos = rng.uniform(-1, 1, size=(n, 9)).astype(np.float32)
        w = rt.bvh_wide_build_host(pos)
        walk_and_check(w["nodes"], w["order"], pos)
        assert len(w["nodes"]) >= 1
    # flat and duplicated geometry: zero-extent axes (cell size irrelevant), identical triangles
    pos = np.zeros((40, 9), dtype=np.float32)
    pos[:, [0, 3, 6]] = rng.uniform(0, 4, size=(40, 3))
    pos[:, [1, 4, 7]] = rng.uniform(0, 4, size=(40, 3))  # all in the plane z = 0
    pos[20:] = pos[:20]
    w = rt.bvh_wide_build_host(pos)
    walk_and_check(w["nodes"], w["order"], pos)
    # huge and tiny coordinates together
    pos = rng.uniform(-1, 1, size=(64, 9)).astype(np.float32)
    pos[:32] *= np.float32(1e6)
    pos[32:] *= np.float32(1e-6)
    w = rt.bvh_wide_build_host(pos)
    walk_and_check(w["nodes"], w["order"], pos)
    w0 = rt.bvh_wide_build_host(np.zeros((0, 9), dtype=np.float32))
    assert len(w0["nodes"]) == 0


def test_wide_tree_slots_follow_the_octants(rt, sg):
    """Slot s is meant to lie towards corner s of its node: on a big random scene the children's centres must agree in sign
    with their slot's corner direction far more often than chance (this is what makes the octant visiting order front to back)."""
    sc = make_scene(sg, dict(kind="room", n_random=5000, seed=3, n_lights=2, n_materials=4, tex_size=0))
    w = rt.bvh_wide_build_host(sc.positions)
    D = decode(w["nodes"])
    agree = total = 0
    for i in range(len(w["nodes"])):
        cell = np.ldexp(1.0, D["e"][i] - 127)
        used = [(s) for s in range(8) if (D["imask"][i] >> s & 1) or ((D["tri_mask"][i] >> (3 * s)) & 7)]
        if len(used) < 2:
            continue
        ctr = np.array([(D["qlo"][i, :, s] + D["qhi"][i, :, s]) * 0.5 * cell for s in used])
        mid = ctr.mean(axis=0)
        for s, c in zip(used, ctr):
            sign = np.array([1.0 if s >> a & 1 else -1.0 for a in range(3)])
            agree += int(((c - mid) * sign > 0).sum())
            total += 3
    assert agree / total > 0.7, agree / total


def test_non_finite_geometry_is_memory_safe(rt):
    """NaN / infinite / extreme coordinates through both host builders: no crash (tools/sanitize_cpu.sh runs this under AddressSanitizer: the
    wide builder's slot assignment once indexed with -1 when every score was NaN), a tree over all triangles comes back. rt_create itself
    refuses non-finite positions (tests/test_gpu_parity.py), these entry points take raw arrays."""
    rng = np.random.default_rng(1)
    for trial in range(12):
        n = int(rng.integers(40, 1500))
        pos = rng.uniform(-1, 1, size=(n, 9)).astype(np.float32)
        k = int(rng.integers(1, max(2, n // 3)))
        idx, col = rng.integers(0, n, size=k), rng.integers(0, 9, size=k)
        pos[idx, col] = [np.nan, np.inf, -np.inf, 3.0e38, -3.0e38, 1e-45][trial % 6]
        b = rt.bvh_build_host(pos)
        w = rt.bvh_wide_build_host(pos)
        assert sorted(w["order"].tolist()) == list(range(n)) and len(w["nodes"]) >= 1
        assert sorted(b["order"].tolist()) == list(range(n))

"""Multi-GPU scenes inside the library (rt_create with RT_ALL_DEVICES / rt_create_on; csrc/rt_group.cpp): replicas, interleaved
blocks, gather on the first GPU. A one-GPU box can run (a) the real RCCL path with G = 1, where RT_BUILD_GROUP_SELF_EXCHANGE routes
the GPU's own blocks through pack -> ncclSend/ncclRecv -> unpack, and (b) G > 1 with repeated ordinals over the peer-copy
rehearsal transport RT_BUILD_GROUP_COPY (RCCL refuses a device twice in one communicator — which is also how RT_ERR_COMM is provoked)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(sg):
    return sg.room_scene(900, seed=51, n_lights=5, n_materials=6, tex_size=16, n_tex_sets=2, alpha_fraction=0.2)


@pytest.fixture(scope="module")
def single(gpu, scene):
    dev = gpu.DeviceScene(scene)
    yield dev
    dev.close()


W, H, SPP = 72, 50, 4  # 3600 pixels: not a multiple of any block size used below -> a partial last block


def test_all_devices_scene_uses_rccl_and_matches_single_gpu(gpu, scene, single, monkeypatch):
    """rt_create(RT_ALL_DEVICES): communicator created inside rt_create (ncclCommInitAll over every visible GPU); with
    RT_BUILD_GROUP_SELF_EXCHANGE the first GPU's own blocks travel through ncclSend/ncclRecv as well, so the RCCL exchange really
    runs even with one GPU. Images must equal the single-GPU render bit for bit (float and rgb8)."""
    grp = gpu.DeviceScene(scene, device=gpu.RT_ALL_DEVICES, build_flags=gpu.RT_BUILD_GROUP_SELF_EXCHANGE)
    try:
        assert grp.n_devices == gpu.device_count() >= 1 and single.n_devices == 1
        want, wst = single.run_raytracer(W, H, SPP, seed=7, counters=True)
        for block in (0, 256, 1000):
            got, gst = grp.run_raytracer(W, H, SPP, seed=7, shard_block=block, counters=True)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), block
            for k in ("samples", "casts", "nodes_visited", "tri_tests", "shaded_hits", "texel_fetches"):
                assert gst[k] == wst[k], k
        img, _ = grp.run_raytracer_rgb8(W, H, SPP, seed=7)
        assert np.array_equal(img, gpu.tonemap(want))
        ref, _ = single.run_raytracer(W, H, 2, rng_mode=gpu.RT_RNG_REFERENCE)
        got, _ = grp.run_raytracer(W, H, 2, rng_mode=gpu.RT_RNG_REFERENCE)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        # the probe entry points are served by the first GPU's replica
        rays = np.random.default_rng(3).normal(size=(500, 6)).astype(np.float32)
        assert np.array_equal(grp.cast_rays(rays)[0], single.cast_rays(rays)[0])
        assert np.array_equal(grp.bvh_info(0)["nodes"], single.bvh_info(0)["nodes"])
        with pytest.raises(gpu.RtError) as e:  # a group shards by itself
            grp.run_raytracer(W, H, 1, shard_index=0, shard_count=2, shard_block=256)
        assert e.value.code == 1
    finally:
        grp.close()


@pytest.mark.parametrize("G", [2, 3, 5])
def test_replicated_render_over_rehearsal_transport(gpu, scene, single, monkeypatch, G):
    """G replicas (all on GPU 0) with one host thread each, interleaved blocks, packed slabs gathered on rank 0 and
    de-interleaved: everything of the multi-GPU flow except the RCCL calls themselves (peer copies stand in for them)."""
    grp = gpu.DeviceScene(scene, device=[0] * G, build_flags=gpu.RT_BUILD_GROUP_COPY)
    try:
        assert grp.n_devices == G
        want, _ = single.run_raytracer(W, H, SPP, seed=9)
        for block in (0, 256, 700):
            got, st = grp.run_raytracer(W, H, SPP, seed=9, shard_block=block)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (G, block)
            assert st["samples"] == W * H * SPP
        img, _ = grp.run_raytracer_rgb8(W, H, SPP, seed=9)
        assert np.array_equal(img, gpu.tonemap(want))
        ref, _ = single.run_raytracer(W, H, 2, rng_mode=gpu.RT_RNG_REFERENCE)
        got, _ = grp.run_raytracer(W, H, 2, rng_mode=gpu.RT_RNG_REFERENCE)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    finally:
        grp.close()


def test_group_with_production_build_and_environment_map(gpu, sg, oracle, monkeypatch):
    """Three replicas of a scene with an environment map, each building the production tree on its device (PLOC + wide collapse): the gathered image
    equals one replica's, and the parity-build group equals the ORACLE (the host half of rt_create, incl. the map's texture view, is shared)."""
    import os

    sc = sg.room_scene(700, seed=52, n_lights=0, n_materials=5, tex_size=8, n_tex_sets=2, open_room=True)
    env = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "envmap", "env.png")
    sc.textures = list(sc.textures) + [gpu.image_decode(env)]
    sc.bg_texture = len(sc.textures) - 1
    orc = oracle.OracleScene(sc)
    ofb, _ = orc.run_raytracer(W, H, SPP, seed=4)
    orc.close()
    grp = gpu.DeviceScene(sc, device=[0, 0, 0], build_flags=gpu.RT_BUILD_GROUP_COPY)
    seen = []
    got, _ = grp.run_raytracer(W, H, SPP, seed=4, shard_block=256, progress=lambda done, total: seen.append((done, total)))
    assert np.array_equal(got.view(np.uint32), ofb.view(np.uint32))
    assert seen == [(1, 3), (2, 3), (3, 3)]  # rt_params.progress on a multi-GPU scene: one report per GPU that finished
    d = np.random.default_rng(8).normal(size=(300, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    one = gpu.DeviceScene(sc)
    assert np.array_equal(grp.bg_at(d).view(np.uint32), one.bg_at(d).view(np.uint32))  # the probe is served by the first replica
    one.close()
    grp.close()
    one = gpu.DeviceScene(sc, device_bvh=True, wide=True)
    grp = gpu.DeviceScene(sc, device=[0, 0, 0], device_bvh=True, wide=True, build_flags=gpu.RT_BUILD_GROUP_COPY)
    want, _ = one.run_raytracer(W, H, SPP, seed=4)
    got, _ = grp.run_raytracer(W, H, SPP, seed=4, shard_block=256)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))  # every replica builds the same tree from the same arrays
    differing = int((got.view(np.uint32) != ofb.view(np.uint32)).any(axis=2).sum())
    assert differing <= 0.01 * W * H, differing  # production contract against the oracle (ties / the reference's pruning quirk only)
    one.close()
    grp.close()


def test_rccl_refusal_is_reported_as_comm_error(gpu, scene, monkeypatch):
    """The same GPU twice in one communicator: ncclCommInitAll refuses -> RT_ERR_COMM with RCCL's message, no crash,
    no half-built scene left behind."""
    with pytest.raises(gpu.RtError) as e:
        gpu.DeviceScene(scene, device=[0, 0])
    assert e.value.code == 7, str(e.value)  # RT_ERR_COMM
    with pytest.raises(gpu.RtError) as e:
        gpu.DeviceScene(scene, device=[0, 99])
    assert e.value.code in (1, 3)  # invalid ordinal: reported by the replica's rt_create


# ------------------------------------------------------------------------------------------------ one process per GPU (bench.py's torchrun flow)
def _hip_rank(rank, world, port, out_path):
    """One rank of the torch.distributed flow: renders ITS interleaved blocks with librt_amd.so (all ranks share GPU 0 here), then the
    same sharding.FramebufferGather bench.py uses assembles the image on rank 0 — over gloo, staged through the host, because RCCL refuses
    one GPU twice; on a multi-GPU node the identical code runs with backend nccl on device tensors."""
    import importlib
    import os
    import sys

    import torch  # before librt_amd.so: one HIP runtime for both
    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rt = importlib.import_module("raytracing-course-hw-public_amd")
    sharding = importlib.import_module("raytracing-course-hw-public_amd.sharding")
    sc = rt.scenegen.room_scene(900, seed=51, n_lights=5, n_materials=6, tex_size=16, n_tex_sets=2, alpha_fraction=0.2)
    dev = rt.DeviceScene(sc, device=0)
    n_pix, block = W * H, 256
    fb = torch.full((n_pix * 3,), -1.0, dtype=torch.float32, device="cuda:0")  # radiance is never negative: -1 marks 'not written'
    img = torch.zeros(n_pix * 3, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()  # RT_FLAG_DEVICE_FB precondition: the fills ran on torch's stream
    dev.run_raytracer(W, H, SPP, seed=7, shard_index=rank, shard_count=world, shard_block=block, device_fb=fb.data_ptr())
    dev.run_raytracer_rgb8(W, H, SPP, seed=7, shard_index=rank, shard_count=world, shard_block=block, device_rgb8=img.data_ptr())
    host = fb.cpu()
    mine = torch.zeros(n_pix, dtype=torch.bool)
    for b0 in range(rank * block, n_pix, world * block):
        mine[b0 : b0 + block] = True
    written = (host.view(-1, 3) != -1.0).any(dim=1)
    assert not written[~mine].any() and written[mine].all()  # exactly this rank's blocks were rendered
    full = sharding.FramebufferGather(n_pix, block, rank, world, torch.device("cpu")).gather(host)
    full8 = sharding.FramebufferGather(n_pix, block, rank, world, torch.device("cpu"), dtype=torch.uint8).gather(img.cpu())
    if rank == 0:
        np.save(out_path, full.numpy().reshape(H, W, 3))
        np.save(out_path + ".rgb8.npy", full8.numpy().reshape(H, W, 3))
    dev.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_torch_distributed_ranks_render_with_the_hip_library(gpu, scene, single, oracle, tmp_path, world):
    """bench.py --gpus N under torch.distributed.run, rehearsed on one GPU: `world` processes, each a rank with its own DeviceScene on GPU 0
    rendering its blocks through the C-ABI, gathered with sharding.FramebufferGather (tests/test_sharding_gloo.py runs the same flow with the
    CPU oracle standing in for the GPU). The assembled float image and the rgb8 image equal the single-GPU render AND the oracle bit for bit.
    Still unverified on hardware: the same exchange over RCCL between different GPUs (no multi-GPU node is available to a round's own runs)."""
    import os

    import torch.multiprocessing as mp

    port = 29600 + (os.getpid() + world) % 2000
    out = str(tmp_path / "full.npy")
    mp.spawn(_hip_rank, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    want, _ = single.run_raytracer(W, H, SPP, seed=7)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    orc = oracle.OracleScene(scene)
    ofb, _ = orc.run_raytracer(W, H, SPP, seed=7)
    orc.close()
    assert np.array_equal(got.view(np.uint32), ofb.view(np.uint32))
    assert np.array_equal(np.load(out + ".rgb8.npy"), gpu.tonemap(want))

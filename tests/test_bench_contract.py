"""bench.py keeps the driver's contract: ONE JSON line with the agreed keys, the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--width", "96", "--height", "64", "--spp", "4",
           "--triangles", "3000", "--tex-size", "16", "--cpu-seconds", "0.5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "Msamples/s" and j["n_gpus"] == 1 and j["steps"] == 1 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["data"] == "synthetic" and j["vs_baseline"] is None and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and abs(j["value"] - 96 * 64 * 4 / (j["ms_per_step"] * 1e3)) / j["value"] < 0.02
    rf = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    # frac follows the contract's formula (ALGORITHMIC bytes / launch time / peak; may exceed 1 on a cache-resident scene) and says so;
    # the HBM-side fractions from the PMC counters stand beside it under their own names, x1 (FETCH_SIZE as counted) and x2 (doubled), null with the reason
    assert rf["frac"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["frac"] == rf["frac_algorithmic"] and "frac_algorithmic" in rf["frac_is"]
    for k in ("frac_is", "frac_algorithmic", "frac_hbm_counters_x1", "frac_hbm_counters_x2", "valu_busy", "lanes_per_valu", "l1_pending_stall_frac", "traffic_fetch_x1", "packet"):
        assert k in rf, k
    assert rf["packet"]["passes"] == 1 and rf["packet"]["mode"] == "auto"
    for k in ("traffic_source", "frac_fetch_x1", "fetch_size_multiplier", "algorithmic_GBps", "algorithmic_over_peak", "binding_roof", "hbm_rate", "hbm_frac",
              "request_rate_Greq_s", "request_roof_Greq_s", "request_frac", "l1_frac", "limiter", "pmc", "peak_measured_read", "peak_measured_source", "kernel",
              "kernel_src_sha16", "launches_per_step", "avg_launch_ms", "algorithmic_bytes_per_launch", "pipeline"):
        assert k in rf, k
    assert len(rf["kernel_src_sha16"]) == 16 and rf["algorithmic_GBps"] > 0
    # a custom (tiny) configuration has no committed profile: traffic is null and the reason is given, never a stale number
    assert j["config"]["workload_id"] == "sponza-custom" and rf["traffic"] is None and rf["frac_hbm_counters_x1"] is None and rf["frac_hbm_counters_x2"] is None and rf["hbm_frac"] is None
    assert "absent" in rf["traffic_source"] and rf["valu_busy"] is None
    assert rf["pmc"]["source"] is None and "absent" in rf["pmc"]["note"] and rf["binding_roof"] is None
    assert rf["peak_measured_read"] and rf["peak_measured_source"].startswith("profiles/")
    for k in ("casts_per_sample", "nodes_per_cast", "tri_tests_per_cast", "algorithmic_bytes_per_sample"):
        assert rf["pipeline"][k] > 0
    # one GPU: the production modes ride on the same line, each with its own record, and are faster than parity or say so
    pr = j["production"]
    for mode in ("global", "wide"):
        assert pr[mode]["value"] > 0 and pr[mode]["roofline"]["pipeline"]["nodes_per_cast"] > 0 and pr[mode]["workload_id"].startswith("sponza-custom")
        cmp_ = pr[mode]["image_vs_parity"]  # the production image against the parity image of the same run, counted
        assert cmp_["pixels"] == 96 * 64 and cmp_["spp"] == 4 and 0 <= cmp_["pixels_beyond_1e-5_relative"] <= cmp_["pixels_differing_from_parity"] <= 0.02 * cmp_["pixels"], cmp_
    assert j["config4"] is None  # custom geometry: no config-4 record
    assert j["extra_workloads"] is None  # custom sizes: no S-10M leg
    assert j["config"]["launcher"] == "single" and j["config"]["ranks_formed"] == 1
    cb = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0


def test_committed_profiles_are_stamped_and_stale_ones_are_never_quoted():
    """The PMC summaries bench.py quotes on the full-size workloads carry the hash of the device sources they were measured
    on. Either they describe the kernels that are shipped (then every figure the bench line copies must be there), or the
    kernels changed after the last profiling pass — then bench.load_profile must refuse them with a reason, so a stale
    traffic figure can never reach a bench line. RT_STRICT_PROFILES=1 (set by the round's final profiling workflow) turns
    staleness itself into a failure. CPU-only check."""
    sys.path.insert(0, ROOT)
    import bench

    sha = bench.kernel_source_hash()
    strict = os.environ.get("RT_STRICT_PROFILES") == "1"
    for wl in ("sponza", "s10m", "sponza-dev-wide", "s10m-dev-wide"):
        for name in ("hbm_traffic", "pmc_wf_extend"):
            path = os.path.join(ROOT, "profiles", f"{bench.PROFILE_ROUND}_{name}_{wl}.json")
            if not os.path.exists(path):  # not profiled (yet) this round: the bench line then carries null + "absent", nothing stale
                assert bench.load_profile(name, wl, sha)[0] is None and not strict, path
                continue
            raw = json.load(open(path))
            assert len(raw.get("kernel_src_sha16", "")) == 16, (wl, name)
            j, why = bench.load_profile(name, wl, sha)
            if raw["kernel_src_sha16"] != sha:
                assert j is None and "stale" in why, (wl, name)
                assert not strict, f"profiles/{bench.PROFILE_ROUND}_{name}_{wl}.json is stale: re-run tools/final_profile.sh"
                continue
            assert j is not None, why
            if name == "hbm_traffic":
                assert j["hbm_bytes_per_launch"] > 0 and j["read_requests_per_launch"] > 0 and j["request_roof_Greq_s"] > 0
            else:
                for k in ("valu_busy", "lanes_per_valu", "wait_any_frac", "l2_hit", "limiter"):
                    assert j[k], k


def test_bench_never_reports_more_gpus_than_took_part(tmp_path):
    """`--gpus N` must not silently run on fewer devices: N > 1 without torchrun takes the in-process group
    (rt_create_on) and insists on N visible GPUs; a WORLD_SIZE that disagrees with --gpus is refused. CPU-only."""
    sys.path.insert(0, ROOT)
    import bench

    assert bench.resolve_launch(1, env={}) == ("single", 0, 0, 1)
    assert bench.resolve_launch(4, env={}, visible_devices=8) == ("group", 0, 0, 1)
    assert bench.resolve_launch(4, env={"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2"}) == ("torchrun", 2, 2, 4)
    with pytest.raises(SystemExit) as e:
        bench.resolve_launch(2, env={}, visible_devices=1)
    assert "only 1 GPU" in str(e.value)
    with pytest.raises(SystemExit):
        bench.resolve_launch(8, env={"WORLD_SIZE": "2"})
    # BASELINE configs: weak = 64 SPP per GPU at EVERY N (one regime per scaling curve; VERDICT r03: auto used to switch to config 4 at 8 GPUs and
    # would have spliced two regimes into the driver's 1 -> 8 curve); strong = config 4 exactly, on request (and as the `config4` record at every N)
    wl = bench.WORKLOADS["sponza"]
    assert bench.resolve_spp("auto", 1, wl, 0) == (64, "weak") and bench.resolve_spp("auto", 4, wl, 0) == (256, "weak")
    assert bench.resolve_spp("auto", 8, wl, 0) == (512, "weak") and bench.resolve_spp("strong", 2, wl, 0) == (1000, "strong")
    render, build = bench.tuning_from_env({"RT_WF_SORT": "0", "RT_WF_PACKET": "1", "RT_WF_MAX_PATHS": "3e6", "RT_PLOC_RADIUS": "4", "RT_DEVICE_BUILDER": "lbvh"})
    assert render == {"sort_mode": 1, "packet_mode": 2, "max_paths": 3000000} and build == {"ploc_radius": 4, "device_builder": 1}
    assert bench.tuning_from_env({}) == ({}, {})
    assert bench.resolve_spp("weak", 8, wl, 0) == (512, "weak")
    # the real command line on this (GPU-less or one-GPU) machine: non-zero exit, the reason on stderr, no JSON line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    if r.returncode == 0:  # a box with >= 2 GPUs really ran it: then the line must say 2
        assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 2
    else:
        assert "GPU(s) visible" in r.stderr and not r.stdout.strip()


def test_library_carries_the_hash_of_its_sources(rt):
    """librt_amd.so is stamped at build time with the sha256 of csrc/device_sources.txt's files (rt_source_stamp); bench.py hashes the
    same list at run time and refuses to measure a library built from other sources. After a build the two must agree, and the
    list must name the kernels, the layouts, the builders, the launch policy and the Makefile (a launch-geometry change
    in rt_scene.cpp or a flag in the Makefile used to keep quoting stale counters)."""
    sys.path.insert(0, ROOT)
    import bench

    assert rt.lib().rt_source_stamp().decode() == bench.kernel_source_hash()
    names = {os.path.basename(f) for f in bench.DEVICE_SOURCES}
    for must in ("rt_wavefront.hip", "rt_wide.hip", "rt_device_lib.h", "rt_device_types.h", "rt_kernels.h", "rt_scene.cpp", "wide_build.cpp", "rt_bvh_device.hip", "Makefile", "rt_devspec.h"):
        assert must in names, must
    for f in bench.DEVICE_SOURCES:
        assert os.path.exists(os.path.join(bench.CSRC, f)), f

#!/usr/bin/env python3
"""BASELINE.md section 3, "cross-check here": the genuine reference binary (oracle/_ref/raytracer_ref = the unmodified
/root/reference/src/main.cpp, recipe oracle/Makefile) and the CPU oracle port (oracle/liboracle.so, what bench.py's
cpu_baseline leg times on the GPU box) timed on the SAME synthetic scene on this container's cores.

Both are measured by two-SPP differencing: wall(spp_hi) - wall(spp_lo) over the added samples, so glTF
load / BVH builds / process start drop out and what is left is the marginal render rate, the same quantity the GPU
metric counts. The oracle renders in reference-RNG + libm mode with all host threads (raytracer.h:636-662), and its PPM
must be byte-identical to the reference binary's at spp_hi (the port is a fair stand-in only while that holds).

    python tools/baseline_crosscheck.py [--triangles 60000] [--size 192] [--spp 8 56] [--out profiles/r03_cpu_baseline_crosscheck.json]

Container only (needs oracle/_ref). Test infrastructure: nothing here is part of the product.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--triangles", type=int, default=60000)
    ap.add_argument("--size", type=int, default=192)
    ap.add_argument("--spp", type=int, nargs=2, default=[8, 56])
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_cpu_baseline_crosscheck.json"))
    args = ap.parse_args()
    import numpy as np

    import oracle

    rt = importlib.import_module("raytracing-course-hw-public_amd")
    assert oracle.have_reference_build(), "build oracle/_ref first (make -C oracle; needs /root/reference)"
    W = H = args.size
    lo, hi = args.spp
    cores = len(os.sched_getaffinity(0))
    results = []
    for label, tex in (("textured", 64), ("untextured", 0)):
        sc = rt.scenegen.room_scene(args.triangles, seed=0x5EED5EED, tex_size=tex, n_tex_sets=4 if tex else 0, n_materials=16, n_lights=16,
                                    light_strength=20.0, alpha_fraction=0.02, offset=0.15,
                                    camera=rt.scenegen.look_camera((-15.0, 4.0, 0.0), yaw_deg=-90.0, yfov=0.9, aspect=W / H))
        with tempfile.TemporaryDirectory() as td:
            gltf = rt.scenegen.write_gltf(sc, os.path.join(td, "scene.gltf"))
            ref_wall, ppm = {}, {}
            for spp in (lo, hi):
                out = os.path.join(td, f"ref_{spp}.ppm")
                t0 = time.perf_counter()
                subprocess.check_call([oracle.REF_BINARY, gltf, str(W), str(H), str(spp), out], stdout=subprocess.DEVNULL)
                ref_wall[spp] = time.perf_counter() - t0
                ppm[spp] = open(out, "rb").read()
            loaded = rt.parse_gltf_scene(gltf, W / H)  # what the reference's loader made of the file: same objects, same order
            orc = oracle.OracleScene(loaded)
            port_wall, port_img = {}, None
            for spp in (lo, hi):
                t0 = time.perf_counter()
                fb, _ = orc.run_raytracer(W, H, spp, rng_mode=rt.RT_RNG_REFERENCE, threads=cores)
                port_wall[spp] = time.perf_counter() - t0
                port_img = fb
            orc.close()
            hdr = f"P6\n{W} {H}\n255\n".encode()
            port_ppm = hdr + oracle.tonemap(port_img).tobytes()
            identical = port_ppm == ppm[hi]
        added = W * H * (hi - lo)
        rec = {
            "scene": f"room_scene({args.triangles} triangles, {label}), {W}x{H}, ray_depth 8, white environment",
            "spp_pair": [lo, hi],
            "reference_binary": {"wall_s": {str(k): round(v, 3) for k, v in ref_wall.items()}, "marginal_Msamples_s": round(added / (ref_wall[hi] - ref_wall[lo]) / 1e6, 4)},
            "oracle_port": {"wall_s": {str(k): round(v, 3) for k, v in port_wall.items()}, "marginal_Msamples_s": round(added / (port_wall[hi] - port_wall[lo]) / 1e6, 4)},
            "ppm_byte_identical_at_spp_hi": bool(identical),
        }
        rec["port_over_reference"] = round(rec["oracle_port"]["marginal_Msamples_s"] / rec["reference_binary"]["marginal_Msamples_s"], 3)
        results.append(rec)
        print(json.dumps(rec), flush=True)
        assert identical, "oracle PPM differs from the reference binary's: the port is not a stand-in"
    out = {"what": "reference binary vs CPU oracle port, marginal render rate by two-SPP differencing (tools/baseline_crosscheck.py)",
           "host": {"cores": cores, "compiler": subprocess.check_output(["g++", "--version"], text=True).splitlines()[0]},
           "results": results}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", os.path.relpath(args.out, ROOT))


if __name__ == "__main__":
    main()

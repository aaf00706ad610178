"""Mutation fuzzing of the host readers (PNG / JPEG / Radiance HDR decoders, glTF and scene-txt loaders) — run it against the sanitised build:

    bash tools/sanitize_cpu.sh          # builds /tmp/rt_asan/librt_amd.so
    ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" \
        RT_AMD_LIB=/tmp/rt_asan/librt_amd.so python tools/fuzz_loaders.py [iterations per seed file]

Every committed fixture is a seed; a mutant has a few random bytes overwritten, inserted or removed, or is truncated. The readers must answer
with RT_OK or an error code: a sanitizer report or a crash ends the process (that is the finding). Prints how many mutants each reader accepted."""
import glob, importlib, os, random, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracing-course-hw-public_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
G = os.path.join(ROOT, "tests", "golden")
seeds = {
    "image": sorted(glob.glob(G + "/png/*.png"))[::4] + sorted(glob.glob(G + "/jpeg/*.jpg"))[::3] + sorted(glob.glob(G + "/envmap/*.hdr")) + [G + "/envmap/env.png"],
    "gltf": [G + "/features/features.gltf"],
    "txt": sorted(glob.glob(G + "/txt/*.txt")),
}
rng = random.Random(1234)


def mutate(b):
    b = bytearray(b)
    k = rng.randrange(5)
    if k == 0 and len(b) > 4:
        del b[rng.randrange(len(b)) :]
    for _ in range(rng.randrange(1, 6)):
        if not b:
            break
        i = rng.randrange(len(b))
        m = rng.randrange(4)
        if m == 0:
            b[i] = rng.randrange(256)
        elif m == 1:
            b[i] ^= 1 << rng.randrange(8)
        elif m == 2:
            b[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 5)))
        else:
            del b[i : i + rng.randrange(1, 5)]
    return bytes(b)


with tempfile.TemporaryDirectory() as td:
    for kind, files in seeds.items():
        ok = bad = 0
        for f in files:
            data = open(f, "rb").read()
            ext = os.path.splitext(f)[1]
            # glTF mutants keep their side files reachable: work inside a copy of the fixture directory
            wd = td
            if kind == "gltf":
                wd = os.path.join(td, "g")
                os.makedirs(wd, exist_ok=True)
                for side in glob.glob(os.path.dirname(f) + "/*"):
                    if not os.path.exists(os.path.join(wd, os.path.basename(side))):
                        open(os.path.join(wd, os.path.basename(side)), "wb").write(open(side, "rb").read())
            for it in range(N * (10 if kind != "image" else 1)):
                p = os.path.join(wd, f"m{ext}")
                if kind == "gltf" and it % 2:  # every other glTF mutant keeps the JSON and damages the binary buffer instead
                    open(p, "wb").write(data)
                    side = os.path.join(os.path.dirname(f), "features.bin")
                    open(os.path.join(wd, "features.bin"), "wb").write(mutate(open(side, "rb").read()))
                else:
                    open(p, "wb").write(mutate(data))
                    if kind == "gltf":
                        open(os.path.join(wd, "features.bin"), "wb").write(open(os.path.join(os.path.dirname(f), "features.bin"), "rb").read())
                try:
                    if kind == "image":
                        rt.image_decode(p)
                    elif kind == "gltf":
                        rt.parse_gltf_scene(p, 1.0).close()
                    else:
                        rt.parse_scene_txt(p).close()
                    ok += 1
                except rt.RtError:
                    bad += 1
        print(f"{kind}: {len(files)} seeds, {ok + bad} mutants, {ok} accepted, {bad} refused with an error code", flush=True)
print("no crash, no sanitizer report")

#!/usr/bin/env bash
# Same CLI as the reference's run.sh (reference run.sh:2, src/main.cpp:17-25):
#   ./run.sh <scene.gltf | scene.txt> <width> <height> <samples> <out.ppm>
# The binary is the re-stated host driver (csrc/host/main.cpp) calling the HIP render loop through the C ABI.
# Environment: RT_DEVICE (GPU ordinal; unset = all visible GPUs), RT_RNG_MODE=device|reference, RT_SEED, RT_VERBOSE=1.
DIR="$(cd "$(dirname "$0")" && pwd)"
BIN="$DIR/raytracing-course-hw-public_amd/csrc/rt_main"
if [ ! -x "$BIN" ]; then
    echo "run.sh: $BIN is not built; run ./build.sh first" >&2
    exit 1
fi
exec "$BIN" "$@"

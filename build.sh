#!/usr/bin/env bash
# Counterpart of the reference's build.sh (cmake Release build): builds librt_amd.so + rt_main with hipcc for gfx950
# and the checker under oracle/. Equivalent to: python -c 'import __graft_entry__ as g; g.build()'
set -e
DIR="$(cd "$(dirname "$0")" && pwd)"
make -C "$DIR/raytracing-course-hw-public_amd/csrc" -j"$(nproc)" all
make -C "$DIR/oracle" all

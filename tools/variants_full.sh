#!/usr/bin/env bash
# tools/variants_full.sh name:"-Dflags" … — like variants.sh, but EVERY object that sees the device data layout (rt_device_types.h) is rebuilt
# with the flags: the kernels, the host and device tree builders and rt_scene.cpp. For layout experiments (record sizes, alignment).
set -e
cd "$(dirname "$0")/../raytracing-course-hw-public_amd/csrc"
mkdir -p variants
DEV="/opt/rocm/bin/hipcc -std=c++20 -O3 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-unused-result --offload-arch=gfx950 -fno-slp-vectorize"
HOST="/opt/rocm/bin/hipcc -std=c++20 -O3 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-unused-result -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  d=variants/$name.d; mkdir -p $d
  for f in rt_kernels rt_wavefront rt_wide rt_bvh_device rt_wide_pack rt_film; do $DEV $flags -c $f.hip -o $d/$f.o & done
  for f in rt_scene wide_build bvh_build rt_group; do $HOST $flags -c $f.cpp -o $d/$f.o & done
  wait
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/$name.so host/film.o host/png_decode.o host/jpeg_decode.o host/hdr_decode.o host/gltf_loader.o host/txt_loader.o \
      $d/bvh_build.o $d/wide_build.o $d/rt_scene.o $d/rt_group.o $d/rt_film.o $d/rt_bvh_device.o $d/rt_wide_pack.o $d/rt_kernels.o $d/rt_wavefront.o $d/rt_wide.o -lz -ldl
  rm -rf $d
  echo "built $name ($flags)"
done

#!/usr/bin/env bash
# tools/variants.sh — build tuning variants of librt_amd.so into csrc/variants/<name>.so:  name:"-Dflags"
# (development aid for kernel tuning; the shipped library is csrc/librt_amd.so; select one with RT_AMD_LIB=<path>)
set -e
cd "$(dirname "$0")/../raytracing-course-hw-public_amd/csrc"
mkdir -p variants
CC="/opt/rocm/bin/hipcc -std=c++20 -O3 -ffp-contract=off -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -fno-slp-vectorize"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  $CC $flags -c rt_kernels.hip -o variants/$name.k.o &
  $CC $flags -c rt_wavefront.hip -o variants/$name.w.o &
  $CC $flags -c rt_wide.hip -o variants/$name.x.o &
  wait
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/$name.so host/film.o host/png_decode.o host/jpeg_decode.o host/hdr_decode.o host/gltf_loader.o host/txt_loader.o bvh_build.o wide_build.o rt_scene.o rt_group.o rt_film.o rt_bvh_device.o rt_wide_pack.o variants/$name.k.o variants/$name.w.o variants/$name.x.o -lz -ldl
  rm -f variants/$name.k.o variants/$name.w.o variants/$name.x.o
  echo "built $name ($flags)"
done

#!/usr/bin/env bash
# tools/sanitize_cpu.sh — the host half of librt_amd.so (loaders, picture decoders, BVH builders, scene preparation, the C-ABI) compiled with
# AddressSanitizer + UndefinedBehaviorSanitizer and the CPU test suite run against it (GPU sanitizers are not available on this pool, and
# the kernels have no host build). The device objects are linked in as built by the Makefile. bench.py's subprocess tests are left out:
# PyTorch's own exception handling trips the preloaded runtime.   bash tools/sanitize_cpu.sh   -> /tmp/rt_asan/run.txt
set -e
cd "$(dirname "$0")/../raytracing-course-hw-public_amd/csrc"
make -s librt_amd.so
O=/tmp/rt_asan; mkdir -p $O
for f in host/film.cpp host/png_decode.cpp host/jpeg_decode.cpp host/hdr_decode.cpp host/gltf_loader.cpp host/txt_loader.cpp bvh_build.cpp wide_build.cpp rt_scene.cpp rt_group.cpp; do
  g++ -std=c++20 -O1 -g -ffp-contract=off -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $f -o $O/$(basename $f .cpp).o
done
g++ -shared -fPIC -fsanitize=address,undefined -o $O/librt_amd.so $O/*.o rt_kernels.o rt_wavefront.o rt_wide.o rt_film.o rt_bvh_device.o -L/opt/rocm/lib -lamdhip64 -lz -ldl -lpthread
cd ../..
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" RT_AMD_LIB=$O/librt_amd.so \
  python -m pytest tests -q -s -m "not gpu" -p no:cacheprovider --ignore=tests/test_bench_contract.py > $O/run.txt 2>&1 || true
grep -n "runtime error\|AddressSanitizer\|SUMMARY" $O/run.txt || echo "no sanitizer finding"
tail -1 $O/run.txt

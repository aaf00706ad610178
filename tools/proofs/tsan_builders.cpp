// tools/proofs/tsan_builders.cpp — the host BVH builders (parallel subtree tasks, csrc/bvh_build.cpp; wide collapse, csrc/wide_build.cpp) under ThreadSanitizer:
//   cd raytracing-course-hw-public_amd/csrc && g++ -std=c++20 -O1 -g -ffp-contract=off -fsanitize=thread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ../../tools/proofs/tsan_builders.cpp bvh_build.cpp wide_build.cpp host/film.cpp -o /tmp/tsan_builders -lpthread && /tmp/tsan_builders
// 600 000 triangles = a dozen concurrent subtree tasks; last run: no report.
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
#include "/root/repo/include/rt_host.h"
int main() {
    const uint32_t n = 600000;
    std::mt19937 g(1);
    std::uniform_real_distribution<float> u(-50, 50), e(-0.5f, 0.5f);
    std::vector<float> pos(9ull * n);
    for (uint32_t i = 0; i < n; ++i) {
        float c[3] = {u(g), u(g) * 0.2f, u(g)};
        for (int k = 0; k < 9; ++k) pos[9ull * i + k] = c[k % 3] + e(g);
    }
    std::vector<uint32_t> subset(n), nodes(10ull * (2 * n + 1)), order(n);
    for (uint32_t i = 0; i < n; ++i) subset[i] = i;
    uint32_t nn = 0, root = 0;
    int rc = rt_bvh_build_host(pos.data(), n, subset.data(), n, &nn, &root, nodes.data(), order.data());
    std::printf("rc %d nodes %u root %u\n", rc, nn, root);
    uint32_t wn = 0, depth = 0; double cost = 0;
    rc = rt_bvh_wide_build_host(pos.data(), n, 1.0f, 0.3f, &wn, &depth, &cost, nullptr, 0, nullptr);
    std::printf("wide rc %d nodes %u depth %u\n", rc, wn, depth);
    return 0;
}

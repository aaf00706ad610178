// stdrand_probe.cpp — known-answer generator for the libstdc++ <random> pieces the reference relies on
// (std::minstd_rand, uniform_real_distribution<float>, uniform_int_distribution<>; raytracer.h:96-98,144,229,
// 358,386,458,479,487). No reference code involved; pins include/rt_devspec.h's rt_minstd_* against the real
// library of this toolchain. TEST INFRASTRUCTURE.
//
//   stdrand_probe real <seed> <n>          n floats from uniform_real_distribution<float>(0,1)
//   stdrand_probe int  <seed> <bound> <n>  n ints from uniform_int_distribution<>(0, bound-1)
//   stdrand_probe range <seed> <a> <b> <n> n floats from uniform_real_distribution<float>(a,b)
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

int main(int argc, char **argv) {
    if (argc < 4)
        return 2;
    std::string mode = argv[1];
    std::minstd_rand rng(std::atoi(argv[2]));
    if (mode == "real") {
        int n = std::atoi(argv[3]);
        for (int i = 0; i < n; ++i) {
            auto dist = std::uniform_real_distribution<float>(0.0f, 1.0f);
            std::printf("%a\n", dist(rng));
        }
    } else if (mode == "int") {
        int bound = std::atoi(argv[3]);
        int n = std::atoi(argv[4]);
        for (int i = 0; i < n; ++i)
            std::printf("%d\n", std::uniform_int_distribution<>(0, bound - 1)(rng));
    } else if (mode == "range") {
        float a = std::strtof(argv[3], nullptr), b = std::strtof(argv[4], nullptr);
        int n = std::atoi(argv[5]);
        for (int i = 0; i < n; ++i) {
            auto dist = std::uniform_real_distribution<float>(a, b);
            std::printf("%a\n", dist(rng));
        }
    }
    return 0;
}

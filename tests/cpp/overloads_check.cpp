// overloads_check.cpp -- the glm and Eigen overloads of include/chad/tsdf.hpp (the reference's tsdf.hpp:68-115) through a compiler, against
// test-only stand-ins for the two headers (tests/cpp/shims/: neither library is in this image).  On a GPU: the same scan through the
// glm, the Eigen and the raw-pointer insert must give the same map.   Usage: overloads_check [points]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "chad/tsdf.hpp"

#if !defined(CHAD_HAS_GLM) || !defined(CHAD_HAS_EIGEN)
#error "the shims were not found: compile with -I tests/cpp/shims"
#endif

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? size_t(std::atol(argv[1])) : 20000;
    std::vector<glm::vec3> g;
    std::vector<Eigen::Vector3f> e;
    std::vector<float> raw;
    unsigned long long s = 88172645463325252ull;
    auto u01 = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return float(double(s >> 11) * (1.0 / 9007199254740992.0)); };
    for (size_t i = 0; i < n; i++) {   // a wall at x = 3 m seen from the origin
        const float y = -2.0f + 4.0f * u01(), z = -1.0f + 2.0f * u01();
        g.emplace_back(3.0f, y, z); e.emplace_back(3.0f, y, z);
        raw.push_back(3.0f); raw.push_back(y); raw.push_back(z);
    }
    try {
        chad::TSDFMap a(0.05f, 0.1f, g, glm::vec3(0.f, 0.f, 0.f));                 // tsdf.hpp:70-73
        chad::TSDFMap b(0.05f, 0.1f, e, Eigen::Vector3f(0.f, 0.f, 0.f));           // tsdf.hpp:95-98
        chad::TSDFMap c;
        c.insert(raw.data(), n, 0.f, 0.f, 0.f);                                    // tsdf.hpp:59
        a.insert(std::vector<glm::vec3>(), glm::vec3(0.f, 0.f, 0.f));              // an empty frame through the overload
        const auto la = a.leaves(), lb = b.leaves(), lc = c.leaves();
        if (la.code != lc.code || lb.code != lc.code || la.weight != lc.weight || lb.weight != lc.weight) return 3;
        if (la.signed_distance != lc.signed_distance || lb.signed_distance != lc.signed_distance) return 4;
        std::printf("ok %zu voxels through each overload\n", lc.code.size());
    } catch (const std::exception& ex) {
        std::printf("error: %s\n", ex.what());
        return 1;
    }
    return 0;
}

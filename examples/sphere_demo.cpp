// sphere_demo.cpp -- the scenario of the reference's src/chad/main.cpp (1 M points on a 5 m sphere, voxel 0.05 m,
// trunc 0.10 m, one insert, save) through the drop-in chad::TSDFMap of include/chad/tsdf.hpp.
// Build: make demo      Run: examples/sphere_demo [points]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "chad/tsdf.hpp"

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000000;
    std::vector<std::array<float, 3>> points(n);
    std::mt19937 gen(420);
    std::uniform_real_distribution<double> dis(-1.0, 1.0);
    for (auto& p : points) {
        double x, y, z, r;
        do { x = dis(gen); y = dis(gen); z = dis(gen); r = std::sqrt(x * x + y * y + z * z); } while (r < 1e-3);
        p = { float(5.0 * x / r), float(5.0 * y / r), float(5.0 * z / r) };
    }
    try {
        chad::TSDFMap map{ 0.05f, 0.1f };
        map.insert(points, { 0.0f, 0.0f, 0.0f });  // first call also loads the code object
        map.sync();
        const auto t0 = std::chrono::steady_clock::now();
        map.insert(points, { 0.0f, 0.0f, 0.0f });
        map.sync();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const auto leaves = map.leaves();
        unsigned long long updates = 0;
        for (auto w : leaves.weight) updates += w;
        std::printf("%zu points: %.3f ms per insert (host points, PCIe copy included), %zu voxels, %llu observations\n", n, ms, leaves.code.size(), updates);
        map.save("mesh.ply");                        // main.cpp:38: the first submap, reconstructed on the GPU
        std::printf("wrote mesh.ply (%zu submap(s) closed)\n", map.closed_submaps());
        // the iterator the reference sketches (tsdf.hpp:120-161): every leaf of the closed submap, position and data, no Morton codes
        size_t count = 0;
        double worst = 0.0;
        for (auto it = map.begin(0); it; ++it, ++count) {
            const chad::TSDFMap::Leaf leaf = *it;
            const double r = std::sqrt(double(leaf.position[0]) * leaf.position[0] + double(leaf.position[1]) * leaf.position[1] + double(leaf.position[2]) * leaf.position[2]);
            // inside the sphere (towards the sensor) the signed distance is positive: r + sd should be about 5 m wherever |sd| < trunc
            if (std::fabs(leaf.signed_distance) < 0.09f) worst = std::max(worst, std::fabs(r + leaf.signed_distance - 5.0));
        }
        std::printf("iterated %zu leaves; |r + sd - 5 m| <= %.3f m\n", count, worst);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

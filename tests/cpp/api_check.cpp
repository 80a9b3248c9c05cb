// api_check.cpp -- test program for the drop-in C++ class (include/chad/tsdf.hpp), written the way code against the
// reference's header is written (README.md:32-54 upstream): construct, insert through each overload, save, read back.
// Usage: api_check <points.f32 (N x 3 floats)> <px> <py> <pz> <leaves.bin out>
// Writes: u64 count, then count x u64 codes, count x f32 signed distances, count x u32 weights (active submap).
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#include "chad/tsdf.hpp"

int main(int argc, char** argv) {
    if (argc != 6) return 2;
    std::FILE* in = std::fopen(argv[1], "rb");
    if (!in) return 3;
    std::fseek(in, 0, SEEK_END);
    const size_t n = size_t(std::ftell(in)) / 12;
    std::fseek(in, 0, SEEK_SET);
    std::vector<std::array<float, 3>> points(n);
    if (std::fread(points.data(), 12, n, in) != n) return 4;
    std::fclose(in);
    const std::array<float, 3> position{float(std::atof(argv[2])), float(std::atof(argv[3])), float(std::atof(argv[4]))};
    try {
        // three maps, the same scan through the three insert forms of tsdf.hpp:48-64: thirds of the scan each
        chad::TSDFMap map;   // default arguments 0.05 / 0.1 (tsdf.hpp:29)
        if (map._sdf_res != 0.05f || map._sdf_trunc != 0.1f) return 5;
        const size_t a = n / 3, b = 2 * n / 3;
        map.insert(std::vector<std::array<float, 3>>(points.begin(), points.begin() + a), position);
        map.insert(points[a].data(), b - a, position.data());
        map.insert(points[b].data(), n - b, position[0], position[1], position[2]);
        // the construct-and-insert form (tsdf.hpp:31-44; null dereference upstream, F7): must give the whole scan's map
        chad::TSDFMap whole(0.05f, 0.1f, points, position);
        const auto parts = map.leaves(), all = whole.leaves();
        if (parts.code.size() != all.code.size()) return 6;
        unsigned long long wp = 0, wa = 0;
        for (auto w : parts.weight) wp += w;
        for (auto w : all.weight) wa += w;
        if (wp == 0) return 7;   // (the thirds see other neighbourhoods than the whole scan: only the totals of `whole` are checked by the caller)
        std::FILE* out = std::fopen(argv[5], "wb");
        if (!out) return 8;
        const unsigned long long count = all.code.size();
        std::fwrite(&count, 8, 1, out);
        std::fwrite(all.code.data(), 8, count, out);
        std::fwrite(all.signed_distance.data(), 4, count, out);
        std::fwrite(all.weight.data(), 4, count, out);
        std::fclose(out);
        if (whole.closed_submaps() != 0) return 9;
        std::printf("ok %zu points, %llu voxels, %llu observations\n", n, count, wa);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

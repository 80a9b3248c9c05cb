// api_check.cpp -- test program for the drop-in C++ class (include/chad/tsdf.hpp), written the way code against the
// reference's header is written (README.md:32-54 upstream): construct, insert through each overload, save, read back.
// Usage: api_check <points.f32 (N x 3 floats)> <px> <py> <pz> <leaves.bin out>
// Writes: u64 count, then count x u64 codes, count x f32 signed distances, count x u32 weights (active submap).
#include <cmath>
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
        // the read-back API the reference sketches (tsdf.hpp:120-161) through the C++ class.  Queries and the raycast look at the
        // ACTIVE submap: the scan's first point lies on the surface, so the field crosses zero there, one range away from the sensor
        const std::array<float, 3>& p0 = points[0];
        const float dx = p0[0] - position[0], dy = p0[1] - position[1], dz = p0[2] - position[2];
        const float range = std::sqrt(dx * dx + dy * dy + dz * dz);
        const auto sd = whole.signed_distance_at({p0}, true);
        const auto hit = whole.raycast({position}, {{dx / range, dy / range, dz / range}}, range + 1.0f);
        if (sd.size() != 1 || hit.size() != 1) return 10;
        if (sd[0] == sd[0] && std::fabs(sd[0]) > 0.05f) return 11;                 // (NaN = an incomplete cell there: allowed)
        if (hit[0] == hit[0] && std::fabs(hit[0] - range) > 0.1f) return 12;
        // save() closes the submap and writes the first one as a triangle mesh (tsdf.cpp:76-86); then its leaves once forwards, once backwards
        whole.save(std::string(argv[5]) + ".ply");
        if (whole.closed_submaps() != 1) return 13;
        unsigned long long forwards = 0, backwards = 0;
        for (auto it = whole.begin(0); it; ++it) forwards++;
        for (auto it = whole.last(0); it; --it) backwards++;
        if (forwards != count || backwards != count) return 14;
        auto it = whole.begin(0);
        it += 5; it -= 2;
        const chad::TSDFMap::Leaf leaf = *it;
        if (!(std::fabs(leaf.signed_distance) <= 0.1f * 1.0001f) || leaf.weight != 255u) return 15;   // Q6: the DAG's weight byte
        std::printf("ok %zu points, %llu voxels, %llu observations\n", n, count, wa);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

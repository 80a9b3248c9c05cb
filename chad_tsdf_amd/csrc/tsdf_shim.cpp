// tsdf_shim.cpp -- chad::TSDFMap (include/chad/tsdf.hpp) over the C ABI; the counterpart of the reference's
// src/chad/tsdf.cpp:26-86.  Converts chad_status failures into std::runtime_error.
#include "chad/tsdf.hpp"

#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "chad_hip.h"

namespace {
void check(chad_status s, const char* what) {
    if (s != CHAD_OK) throw std::runtime_error(std::string(what) + ": " + chad_last_error());
}
}  // namespace

namespace chad {

TSDFMap::TSDFMap(float sdf_res, float sdf_trunc) : _sdf_res(sdf_res), _sdf_trunc(sdf_trunc) {
    chad_options opt{};
    opt.struct_size = sizeof(opt);
    opt.device = -1;
    const char* verbose = std::getenv("CHAD_VERBOSE");
    if (verbose && verbose[0] && verbose[0] != '0') opt.flags |= CHAD_FLAG_VERBOSE;
    if (const char* blocks = std::getenv("CHAD_MAX_BLOCKS")) opt.max_blocks = std::strtoull(blocks, nullptr, 10);
    check(chad_map_create(sdf_res, sdf_trunc, &opt, &_impl), "chad::TSDFMap");
}

TSDFMap::~TSDFMap() { chad_map_destroy(_impl); }

void TSDFMap::insert(const float* points_p, size_t points_count, const float* position_p) {
    check(chad_map_insert(_impl, points_p, points_count, position_p), "chad::TSDFMap::insert");
}

void TSDFMap::sync() { check(chad_map_sync(_impl), "chad::TSDFMap::sync"); }

size_t TSDFMap::closed_submaps() const {
    chad_stats st{};
    check(chad_map_stats(_impl, &st), "chad::TSDFMap::closed_submaps");
    return size_t(st.closed_submaps);
}

TSDFMap::Leaves TSDFMap::leaves(int submap) const {
    Leaves out;
    size_t n = 0;
    check(chad_map_leaf_count(_impl, submap, &n), "chad::TSDFMap::leaves");
    out.code.resize(n); out.signed_distance.resize(n); out.weight.resize(n);
    check(chad_map_export_leaves(_impl, submap, out.code.data(), out.signed_distance.data(), out.weight.data(), n, &n), "chad::TSDFMap::leaves");
    return out;
}

void TSDFMap::save(const std::string& filename) {
    // tsdf.cpp:78-85: the active submap is closed if it holds a pose, then the FIRST submap is written.
    check(chad_map_close_submap(_impl), "chad::TSDFMap::save");
    check(chad_map_finalize(_impl), "chad::TSDFMap::save");  // Submap::finalize of every submap closed so far (tsdf.cpp:53,79)
    if (closed_submaps() == 0) throw std::runtime_error("chad::TSDFMap::save: nothing was inserted");
    const Leaves first = leaves(0);
    std::FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) throw std::runtime_error("chad::TSDFMap::save: cannot open " + filename);
    // binary PLY: one vertex per voxel at its lower corner (the point the reference measures from, octree.hpp:157)
    std::fprintf(f, "ply\nformat binary_little_endian 1.0\ncomment chad_tsdf voxels res %g trunc %g\nelement vertex %zu\n"
                    "property float x\nproperty float y\nproperty float z\nproperty float signed_distance\nproperty uint weight\nend_header\n",
                 double(_sdf_res), double(_sdf_trunc), first.code.size());
    auto compact = [](uint64_t v) {  // inverse of the 21-bit interleave (morton.hpp:29-37)
        v &= 0x1249249249249249ull;
        v = (v | (v >> 2)) & 0x10c30c30c30c30c3ull;
        v = (v | (v >> 4)) & 0x100f00f00f00f00full;
        v = (v | (v >> 8)) & 0x001f0000ff0000ffull;
        v = (v | (v >> 16)) & 0x001f00000000ffffull;
        v = (v | (v >> 32)) & 0x1fffffull;
        return int32_t(uint32_t(v) - (1u << 20));
    };
    for (size_t i = 0; i < first.code.size(); i++) {
        const uint64_t mc = first.code[i];
        const float rec[4] = { float(compact(mc)) * _sdf_res, float(compact(mc >> 1)) * _sdf_res, float(compact(mc >> 2)) * _sdf_res, first.signed_distance[i] };
        std::fwrite(rec, 4, 4, f);
        std::fwrite(&first.weight[i], 4, 1, f);
    }
    std::fclose(f);
}

}  // namespace chad

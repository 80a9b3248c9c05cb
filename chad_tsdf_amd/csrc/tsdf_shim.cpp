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
    // tsdf.cpp:78-85: the active submap is closed if it holds a pose, then the FIRST submap is reconstructed and written.
    check(chad_map_close_submap(_impl), "chad::TSDFMap::save");
    check(chad_map_finalize(_impl), "chad::TSDFMap::save");  // Submap::finalize of every submap closed so far (tsdf.cpp:53,79)
    if (closed_submaps() == 0) throw std::runtime_error("chad::TSDFMap::save: nothing was inserted");
    check(chad_map_save_mesh(_impl, 0, filename.c_str()), "chad::TSDFMap::save");   // detail::reconstruct(*_submaps.front(), ...), tsdf.cpp:85
}

// ---- the iterator the reference sketches (tsdf.hpp:120-161, tsdf.cpp:88-159) ----
TSDFMap::iterator::iterator(chad_iter* it) : _it(it) {}
TSDFMap::iterator::iterator(iterator&& other) noexcept : _it(other._it) { other._it = nullptr; }
TSDFMap::iterator& TSDFMap::iterator::operator=(iterator&& other) noexcept {
    if (this != &other) { chad_iter_destroy(_it); _it = other._it; other._it = nullptr; }
    return *this;
}
TSDFMap::iterator::~iterator() { chad_iter_destroy(_it); }
bool TSDFMap::iterator::valid() const { return _it && chad_iter_valid(_it) != 0; }
TSDFMap::iterator& TSDFMap::iterator::operator++() { check(chad_iter_advance(_it, 1, nullptr), "chad::TSDFMap::iterator++"); return *this; }
TSDFMap::iterator& TSDFMap::iterator::operator--() { check(chad_iter_advance(_it, -1, nullptr), "chad::TSDFMap::iterator--"); return *this; }
TSDFMap::iterator& TSDFMap::iterator::operator+=(int64_t n) { check(chad_iter_advance(_it, n, nullptr), "chad::TSDFMap::iterator+="); return *this; }
TSDFMap::iterator& TSDFMap::iterator::operator-=(int64_t n) { check(chad_iter_advance(_it, -n, nullptr), "chad::TSDFMap::iterator-="); return *this; }
TSDFMap::Leaf TSDFMap::iterator::operator*() const {
    chad_leaf l{};
    check(chad_iter_get(_it, &l), "chad::TSDFMap::iterator*");
    return Leaf{{l.x, l.y, l.z}, l.signed_distance, l.weight};
}
TSDFMap::iterator TSDFMap::begin(uint32_t submap) const {
    chad_iter* it = nullptr;
    check(chad_map_iter_begin(_impl, int32_t(submap), &it), "chad::TSDFMap::begin");   // throws on a faulty submap, as tsdf.cpp:106-110 does on a faulty root
    return iterator(it);
}
TSDFMap::iterator TSDFMap::last(uint32_t submap) const {
    iterator it = begin(submap);
    check(chad_iter_seek(it._it, 1), "chad::TSDFMap::last");
    return it;
}

std::vector<float> TSDFMap::signed_distance_at(const std::vector<std::array<float, 3>>& points, bool trilinear) const {
    std::vector<float> sd(points.size());
    check(chad_map_query_sd(_impl, points.empty() ? nullptr : points.front().data(), points.size(), trilinear ? CHAD_QUERY_TRILINEAR : CHAD_QUERY_LEAF, sd.data(), nullptr),
          "chad::TSDFMap::signed_distance_at");
    return sd;
}
std::vector<float> TSDFMap::raycast(const std::vector<std::array<float, 3>>& origins, const std::vector<std::array<float, 3>>& directions, float max_range) const {
    if (origins.size() != directions.size()) throw std::runtime_error("chad::TSDFMap::raycast: origins and directions differ in size");
    std::vector<float> t(origins.size());
    check(chad_map_raycast(_impl, origins.empty() ? nullptr : origins.front().data(), directions.empty() ? nullptr : directions.front().data(), origins.size(), max_range, t.data()),
          "chad::TSDFMap::raycast");
    return t;
}
void TSDFMap::attach(chad_comm* comm) { check(chad_map_attach_comm(_impl, comm), "chad::TSDFMap::attach"); }

}  // namespace chad

// chad_oracle.cpp -- TEST INFRASTRUCTURE ONLY (the parity checker and the CPU baseline).
//
// A single-threaded CPU restatement of the reference's chad::TSDFMap::insert() path, written from
// the behaviour of /root/reference (M2-TE/chad_tsdf @ 2025-07-25).  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library; the product (libchad_hip.so) never does.
//
// PARITY UNPINNED: upstream ships no tests, fixtures or golden vectors for this path,
// and the reference cannot be built in this image (its CMake fetches glm 1.0.1, gtl v1.2.0,
// libmorton v0.2.12 and fmt 11.2.0 from the network; none is vendored or installed).  The third-party
// pieces on the path are restated from their published definitions:
//   * libmorton v0.2.12 morton3D_64_encode(x,y,z): x -> bits 0,3,6,.., y -> bits 1,4,7,.., z -> bits 2,5,8,..,
//     21 bits per axis (BMI2 pdep masks 0x1249249249249249 / <<1 / <<2);
//   * glm 1.0.1 generic (non-SIMD) vec3 code: dot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z (left to right),
//     normalize(v) = v * (1 / sqrt(dot(v,v))), distance = sqrt(dot(d,d)), floor/abs/sign component-wise;
//   * gtl::flat_hash_map only affects speed (results depend on insertion order, not on the hash).
// The only external anchors are the scalar summaries SURVEY.md section 8(c) G7 records from a reference run
// (distinct voxels, sum of weights, block count); tests/test_oracle_survey_pins.py checks them.
//
// Arithmetic is strict IEEE (build with -ffp-contract=off, no -ffast-math) so that voxel sets are
// well defined; upstream's default -ffast-math never was.
//
// Deviations from the reference, all where the reference has undefined behaviour:
//   D1 (Q3) normals.hpp:100-101 reads one element past the end while grouping the last sorted point;
//      here that comparison counts as a mismatch, so the last point always gets the fallback normal.
//   D2 (Q5) a point whose ray direction normalize(p - position) is not finite (the point equals the sensor
//      position, has a non-finite coordinate, or overflows) makes octree.hpp:92-97 convert NaN to int, and a
//      point beyond 2^30 voxels overflows the int conversion of morton.hpp:73; such points are dropped
//      before Morton encoding.
//   D3 (Q2) std::sort leaves the order of equal codes unspecified (morton.hpp:89); here ties are
//      ordered by ascending input index, which is one of the orders std::sort may produce.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <vector>
#if defined(__BMI2__)
#include <immintrin.h>
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// Morton code: morton.hpp:21-37 over libmorton's 3D 64-bit interleave.
inline uint64_t spread21(uint64_t v) {
    v &= 0x1fffffULL;
    v = (v | (v << 32)) & 0x001f00000000ffffULL;
    v = (v | (v << 16)) & 0x001f0000ff0000ffULL;
    v = (v | (v << 8)) & 0x100f00f00f00f00fULL;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ULL;
    v = (v | (v << 2)) & 0x1249249249249249ULL;
    return v;
}
inline uint32_t compact21(uint64_t v) {
    v &= 0x1249249249249249ULL;
    v = (v | (v >> 2)) & 0x10c30c30c30c30c3ULL;
    v = (v | (v >> 4)) & 0x100f00f00f00f00fULL;
    v = (v | (v >> 8)) & 0x001f0000ff0000ffULL;
    v = (v | (v >> 16)) & 0x001f00000000ffffULL;
    v = (v | (v >> 32)) & 0x1fffffULL;
    return uint32_t(v);
}
#if defined(__BMI2__)
inline uint64_t morton_encode(int32_t vx, int32_t vy, int32_t vz) {
    // morton.hpp:24-27: bias by 2^20 in uint32 arithmetic, then keep 21 bits per axis
    uint32_t x = (1u << 20) + uint32_t(vx), y = (1u << 20) + uint32_t(vy), z = (1u << 20) + uint32_t(vz);
    return _pdep_u64(x, 0x1249249249249249ULL) | _pdep_u64(y, 0x2492492492492492ULL) | _pdep_u64(z, 0x4924924924924924ULL);
}
inline void morton_decode(uint64_t mc, int32_t& vx, int32_t& vy, int32_t& vz) {
    // morton.hpp:30-36
    vx = int32_t(uint32_t(_pext_u64(mc, 0x1249249249249249ULL)) - (1u << 20));
    vy = int32_t(uint32_t(_pext_u64(mc, 0x2492492492492492ULL)) - (1u << 20));
    vz = int32_t(uint32_t(_pext_u64(mc, 0x4924924924924924ULL)) - (1u << 20));
}
#else
inline uint64_t morton_encode(int32_t vx, int32_t vy, int32_t vz) {
    uint32_t x = (1u << 20) + uint32_t(vx), y = (1u << 20) + uint32_t(vy), z = (1u << 20) + uint32_t(vz);
    return spread21(x) | (spread21(y) << 1) | (spread21(z) << 2);
}
inline void morton_decode(uint64_t mc, int32_t& vx, int32_t& vy, int32_t& vz) {
    vx = int32_t(compact21(mc) - (1u << 20));
    vy = int32_t(compact21(mc >> 1) - (1u << 20));
    vz = int32_t(compact21(mc >> 2) - (1u << 20));
}
#endif

struct V3 { float x, y, z; };
struct D3 { double x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// morton.hpp:58 -- pair<vec3, MortonCode> is 24 bytes with 4 bytes of padding; the padding carries the
// input index here so that the tie order (D3) can be made deterministic at no extra memory traffic.
struct PointMc { V3 p; uint32_t idx; uint64_t mc; };
static_assert(sizeof(PointMc) == 24, "same footprint as the reference's sort element");

inline bool point_usable(const float* p, const float* pos, float recip) {  // D2
    const V3 d = normalize(V3{p[0], p[1], p[2]} - V3{pos[0], pos[1], pos[2]});
    const float lim = 1073741824.0f;  // 2^30 voxels
    return std::isfinite(d.x) && std::isfinite(d.y) && std::isfinite(d.z) &&
           std::fabs(p[0] * recip) < lim && std::fabs(p[1] * recip) < lim && std::fabs(p[2] * recip) < lim;
}

// morton.hpp:59-80
void calc_morton_vector(const float* xyz, size_t n, const float pos[3], float sdf_res, std::vector<PointMc>& out) {
    const float recip = float(1.0 / double(sdf_res));  // morton.hpp:63
    out.clear();
    out.reserve(n);
    for (size_t i = 0; i < n; i++) {
        const float* p = xyz + 3 * i;
        if (!point_usable(p, pos, recip)) continue;  // D2
        float fx = std::floor(p[0] * recip), fy = std::floor(p[1] * recip), fz = std::floor(p[2] * recip);  // :71
        out.push_back({{p[0], p[1], p[2]}, uint32_t(i), morton_encode(int32_t(fx), int32_t(fy), int32_t(fz))});  // :73
    }
}

// morton.hpp:81-102 (descending; ties by input index, D3)
void sort_morton_vector(std::vector<PointMc>& v, std::vector<V3>& sorted_points) {
    std::sort(v.begin(), v.end(), [](const PointMc& a, const PointMc& b) {
        return a.mc > b.mc || (a.mc == b.mc && a.idx < b.idx);
    });
    sorted_points.clear();
    sorted_points.reserve(v.size());
    for (const auto& e : v) sorted_points.push_back(e.p);
}

// normals.hpp:10-80 ("plane from points 2")
V3 estimate_normal(const PointMc* beg, const PointMc* end) {
    D3 c{0, 0, 0};
    for (const PointMc* it = beg; it != end; ++it) { c.x += double(it->p.x); c.y += double(it->p.y); c.z += double(it->p.z); }
    const double recip = 1.0 / double(end - beg);
    c.x *= recip; c.y *= recip; c.z *= recip;
    double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    for (const PointMc* it = beg; it != end; ++it) {
        double rx = double(it->p.x) - c.x, ry = double(it->p.y) - c.y, rz = double(it->p.z) - c.z;
        xx += rx * rx; xy += rx * ry; xz += rx * rz; yy += ry * ry; yz += ry * rz; zz += rz * rz;
    }
    xx *= recip; xy *= recip; xz *= recip; yy *= recip; yz *= recip; zz *= recip;
    D3 wd{0, 0, 0};
    auto accumulate = [&wd](D3 axis, double det) {
        double weight = det * det;
        if (dot(wd, axis) < 0.0) weight = -weight;
        wd.x += axis.x * weight; wd.y += axis.y * weight; wd.z += axis.z * weight;
    };
    { double det = yy * zz - yz * yz; accumulate({det, xz * yz - xy * zz, xy * yz - xz * yy}, det); }  // :42-52
    { double det = xx * zz - xz * xz; accumulate({xz * yz - xy * zz, det, xy * xz - yz * xx}, det); }  // :54-64
    { double det = xx * yy - xy * xy; accumulate({xy * yz - xz * yy, xy * xz - yz * xx, det}, det); }  // :66-76
    const double inv = 1.0 / std::sqrt(dot(wd, wd));  // glm::normalize(dvec3)
    return {float(wd.x * inv), float(wd.y * inv), float(wd.z * inv)};  // :79
}

// normals.hpp:81-148.  group_start (optional) gets 1 at the first element of every neighbourhood.
void estimate_normals(const std::vector<PointMc>& v, V3 pos, std::vector<V3>& normals, uint8_t* group_start) {
    const size_t n = v.size();
    normals.resize(n);
    constexpr size_t min_points = 8;  // :88
    for (size_t i = 0; i < n;) {
        size_t end = i + 1;
        for (unsigned depth = 0; depth < 3; depth++) {  // :94
            const uint64_t mask = ~uint64_t(0) << (depth * 3);
            const uint64_t key = v[i].mc & mask;
            // :100 -- the guard stops one short of the last element; at end == n the reference reads out of
            // bounds (D1): treated as a mismatch.
            while (end != n - 1 && end < n) {
                if ((v[end].mc & mask) == key) end++;
                else break;
            }
            if (end - i >= min_points) break;
        }
        const size_t size = end - i;
        if (group_start) group_start[i] = 1;
        if (size >= min_points) {
            V3 nrm = estimate_normal(&v[i], &v[end]);
            if (dot(nrm, normalize(pos - v[i].p)) < 0.0f) nrm = {-nrm.x, -nrm.y, -nrm.z};  // :117-118
            for (size_t j = i; j < end; j++) normals[j] = nrm;
        } else {
            for (size_t j = i; j < end; j++) normals[j] = normalize(pos - v[j].p);  // :127-134
        }
        i += size;
    }
}

// ---------------------------------------------------------------------------------------------
// octree.hpp:86-152 -- voxel traversal of the segment [p - trunc*dir, p + trunc*dir].
struct I3 { int32_t x, y, z; };
inline int32_t isign(int32_t v) { return (v > 0) - (v < 0); }

template <class Emit>
inline void walk_ray(V3 p, V3 pos, float res, float res_recip, float trunc, Emit&& emit) {
    const V3 dir = normalize(p - pos);                                     // :92
    const V3 dir_recip{1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};           // :93
    const V3 start = p - dir * trunc, fin = p + dir * trunc;                // :94-95
    const I3 v0{int32_t(std::floor(start.x * res_recip)), int32_t(std::floor(start.y * res_recip)), int32_t(std::floor(start.z * res_recip))};
    const I3 v1{int32_t(std::floor(fin.x * res_recip)), int32_t(std::floor(fin.y * res_recip)), int32_t(std::floor(fin.z * res_recip))};
    const I3 step{isign(v1.x - v0.x), isign(v1.y - v0.y), isign(v1.z - v0.z)};  // :100
    const V3 delta{std::fabs(res * dir_recip.x), std::fabs(res * dir_recip.y), std::fabs(res * dir_recip.z)};  // :102
    V3 tmax;
    auto boundary = [&](int32_t s, float st) {  // :106-116
        if (s < 0) return res * std::floor(st * res_recip);
        if (s > 0) return res * std::ceil(st * res_recip);
        return std::numeric_limits<float>::max();
    };
    tmax.x = boundary(step.x, start.x); tmax.y = boundary(step.y, start.y); tmax.z = boundary(step.z, start.z);
    tmax = tmax - start;                                                                   // :117
    tmax = {std::fabs(tmax.x * dir_recip.x), std::fabs(tmax.y * dir_recip.y), std::fabs(tmax.z * dir_recip.z)};  // :118
    I3 cur = v0;
    emit(cur);  // :122
    for (;;) {  // :125-152
        if (tmax.x < tmax.y) {
            if (tmax.x < tmax.z) { cur.x += step.x; tmax.x += delta.x; if (cur.x == v1.x + step.x) break; }
            else                 { cur.z += step.z; tmax.z += delta.z; if (cur.z == v1.z + step.z) break; }
        } else {
            if (tmax.y < tmax.z) { cur.y += step.y; tmax.y += delta.y; if (cur.y == v1.y + step.y) break; }
            else                 { cur.z += step.z; tmax.z += delta.z; if (cur.z == v1.z + step.z) break; }
        }
        emit(cur);
    }
}

// The same traversal rule (octree.hpp:100-152) on the segment from `from` to `to`, direction `dir` (= normalize(to - from) up to
// rounding; passed in so that it is bit for bit the ray's).  Used by the space-carving EXTENSION below: upstream has no such code
// (README.md:59-60 lists it as not done).
template <class Emit>
inline void walk_segment(V3 from, V3 to, V3 dir, float res, float res_recip, Emit&& emit) {
    const V3 dir_recip{1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
    const I3 v0{int32_t(std::floor(from.x * res_recip)), int32_t(std::floor(from.y * res_recip)), int32_t(std::floor(from.z * res_recip))};
    const I3 v1{int32_t(std::floor(to.x * res_recip)), int32_t(std::floor(to.y * res_recip)), int32_t(std::floor(to.z * res_recip))};
    const I3 step{isign(v1.x - v0.x), isign(v1.y - v0.y), isign(v1.z - v0.z)};
    const V3 delta{std::fabs(res * dir_recip.x), std::fabs(res * dir_recip.y), std::fabs(res * dir_recip.z)};
    V3 tmax;
    auto boundary = [&](int32_t s, float st) {
        if (s < 0) return res * std::floor(st * res_recip);
        if (s > 0) return res * std::ceil(st * res_recip);
        return std::numeric_limits<float>::max();
    };
    tmax.x = boundary(step.x, from.x); tmax.y = boundary(step.y, from.y); tmax.z = boundary(step.z, from.z);
    tmax = tmax - from;
    tmax = {std::fabs(tmax.x * dir_recip.x), std::fabs(tmax.y * dir_recip.y), std::fabs(tmax.z * dir_recip.z)};
    I3 cur = v0;
    int64_t budget = int64_t(std::abs(int64_t(v1.x) - v0.x)) + std::abs(int64_t(v1.y) - v0.y) + std::abs(int64_t(v1.z) - v0.z);
    emit(cur);
    for (;;) {
        if (tmax.x < tmax.y) {
            if (tmax.x < tmax.z) { cur.x += step.x; tmax.x += delta.x; if (cur.x == v1.x + step.x) break; }
            else                 { cur.z += step.z; tmax.z += delta.z; if (cur.z == v1.z + step.z) break; }
        } else {
            if (tmax.y < tmax.z) { cur.y += step.y; tmax.y += delta.y; if (cur.y == v1.y + step.y) break; }
            else                 { cur.z += step.z; tmax.z += delta.z; if (cur.z == v1.z + step.z) break; }
        }
        if (--budget < 0) break;   // (every step moves one axis towards v1)
        emit(cur);
    }
}

// ---------------------------------------------------------------------------------------------
// octree.hpp:12-78 -- pointer octree of depth 20 + leaves, with a hash shortcut at 4x4x4-voxel granularity.
struct Leaf { float sd; uint32_t w; };
struct Node { uint32_t child[8]; };

struct BlockLookup {  // stands in for gtl::flat_hash_map<MortonCode, Node*> (octree.hpp:187)
    std::vector<uint64_t> keys;
    std::vector<uint32_t> vals;
    size_t mask = 0, count = 0;
    static constexpr uint64_t EMPTY = ~uint64_t(0);
    BlockLookup() { rehash(1 << 12); }
    void rehash(size_t cap) {
        std::vector<uint64_t> ok; ok.swap(keys);
        std::vector<uint32_t> ov; ov.swap(vals);
        keys.assign(cap, EMPTY); vals.assign(cap, 0); mask = cap - 1; count = 0;
        for (size_t i = 0; i < ok.size(); i++) if (ok[i] != EMPTY) *find_or_insert(ok[i]).first = ov[i], void();
    }
    static inline size_t hash(uint64_t k) { k >>= 6; k *= 0x9E3779B97F4A7C15ULL; return size_t(k >> 20); }
    // returns (value slot, inserted)
    std::pair<uint32_t*, bool> find_or_insert(uint64_t key) {
        size_t i = hash(key) & mask;
        for (;; i = (i + 1) & mask) {
            if (keys[i] == key) return {&vals[i], false};
            if (keys[i] == EMPTY) {
                if ((count + 1) * 2 > mask + 1) { rehash((mask + 1) * 2); return find_or_insert(key); }
                keys[i] = key; count++;
                return {&vals[i], true};
            }
        }
    }
    void clear() { std::fill(keys.begin(), keys.end(), EMPTY); count = 0; }
};

struct Octree {
    std::vector<Node> nodes;
    std::vector<Leaf> leaves;
    std::vector<uint64_t> leaf_mc;  // oracle-only: the code of each leaf, for export without a tree walk
    std::vector<double> leaf_sum;   // oracle-only: the f64 sum of the leaf's clamped distances (what the f32 running mean of
                                    // octree.hpp:161-163 approximates), to tell the reference's own rounding drift from a port's error
    BlockLookup lookup;
    Octree() { clear(); }
    void clear() {  // octree.hpp:22-29
        lookup.clear();
        nodes.clear(); leaves.clear(); leaf_mc.clear(); leaf_sum.clear();
        nodes.push_back(Node{});
        leaves.push_back(Leaf{0.0f, 0});
        leaf_mc.push_back(0);
        leaf_sum.push_back(0.0);
    }
    Leaf& find_or_create(uint64_t mc) {  // octree.hpp:31-78
        constexpr uint32_t lookup_depth = 18;
        constexpr uint64_t lookup_mask = 0xffffffffffffffc0ULL;  // :33-35
        auto [val, inserted] = lookup.find_or_insert(mc & lookup_mask);
        uint32_t depth = 0, node = 0;
        if (!inserted) { depth = lookup_depth + 1; node = *val; }
        while (depth < 20) {
            const uint64_t shift = uint64_t(20 - depth) * 3;
            const uint32_t ci = uint32_t(mc >> shift) & 7u;
            uint32_t child = nodes[node].child[ci];
            if (child == 0) {
                child = uint32_t(nodes.size());
                nodes.push_back(Node{});
                nodes[node].child[ci] = child;
                if (inserted && depth == lookup_depth) *val = child;  // :57-59 (cache the depth-19 node)
            }
            node = child;
            depth++;
        }
        const uint32_t li = uint32_t(mc) & 7u;
        uint32_t leaf = nodes[node].child[li];
        if (leaf == 0) {
            leaf = uint32_t(leaves.size());
            leaves.push_back(Leaf{0.0f, 0});
            leaf_mc.push_back(mc);
            leaf_sum.push_back(0.0);
            nodes[node].child[li] = leaf;
        }
        return leaves[leaf];
    }
    // read-only: the leaf of `mc`, or null (a walk from the root, nothing created)
    Leaf* find(uint64_t mc) {
        uint32_t node = 0;
        for (uint32_t depth = 0; depth < 20; depth++) {
            node = nodes[node].child[uint32_t(mc >> (uint64_t(20 - depth) * 3)) & 7u];
            if (node == 0) return nullptr;
        }
        const uint32_t leaf = nodes[node].child[uint32_t(mc) & 7u];
        return leaf ? &leaves[leaf] : nullptr;
    }
};

// ---------------------------------------------------------------------------------------------
// The step right behind the path: Submap::finalize (submap.hpp:10-106) folds the pointer octree into the map-wide
// DAG of NodeLevels (levels.hpp).  Results depend on insertion order only (addresses are handed out in order of
// first occurrence); the gtl hash sets are stood in for by std::unordered_map keyed on the node's content.
struct NodeKey {
    uint32_t w[9]; uint32_t n;  // mask + children, n words
    bool operator==(const NodeKey& o) const { return n == o.n && std::memcmp(w, o.w, n * 4) == 0; }
};
struct NodeKeyHash {
    size_t operator()(const NodeKey& k) const {
        uint64_t h = 0x9E3779B97F4A7C15ULL;
        for (uint32_t i = 0; i < k.n; i++) { h ^= k.w[i]; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 29; }
        return size_t(h);
    }
};
struct NodeLevel {  // levels.hpp:8-94
    std::vector<uint32_t> raw{0};   // :52-53 (address 0 is reserved)
    uint32_t uniques = 0, dupes = 0;
    std::unordered_map<NodeKey, uint32_t, NodeKeyHash> set;
    uint32_t add(const uint32_t children[8]) {  // :56-86
        NodeKey k{}; k.n = 1;
        for (uint32_t i = 0; i < 8; i++) if (children[i]) { k.w[k.n++] = children[i]; k.w[0] |= 1u << i; }
        const uint32_t addr = uint32_t(raw.size());
        auto [it, fresh] = set.try_emplace(k, addr);
        if (!fresh) { dupes++; return it->second; }
        uniques++;
        raw.insert(raw.end(), k.w, k.w + k.n);
        return addr;
    }
};
struct LeafClusterLevel {  // levels.hpp:95-139
    std::vector<uint64_t> raw{0};   // :118-119
    uint32_t uniques = 0, dupes = 0;
    std::unordered_map<uint64_t, uint32_t> set;
    uint32_t add(uint64_t value) {  // :122-134
        auto [it, fresh] = set.try_emplace(value, uniques + 1);
        if (!fresh) { dupes++; return it->second; }
        uniques++;
        raw.push_back(value);
        return uniques;
    }
};
struct NodeLevels {  // levels.hpp:141-199
    static constexpr uint32_t MAX_DEPTH = 20;
    NodeLevel nodes[MAX_DEPTH];
    LeafClusterLevel leaf_clusters;
};

// cluster.hpp:13-26: 8 bits per leaf, signed distance normalised by the truncation distance
inline uint64_t cluster_sd_bits(float signed_distance, float trunc_recip) {
    float sd = std::min(std::max(signed_distance * trunc_recip, -1.0f), 1.0f);
    sd = sd * 127.0f + 127.0f;
    return uint64_t(sd);
}

// submap.hpp:10-106 -- depth-first walk of the pointer octree in child order (= ascending Morton order); every octree
// node whose children are leaves becomes a pair of leaf clusters (signed distances, weights), every node above it a
// pair of DAG nodes, added child-first, the signed-distance one before the weight one.
// Q6 (submap.hpp:93): std::max<uint8_t>(w, 255) makes every stored weight 255, which is also the "empty" marker, so
// every weight cluster is all ones.  saturate_weights = true stores min(w, 254) instead (a documented fix).
void finalize(const Octree& oct, NodeLevels& lv, float trunc, bool saturate_weights, uint32_t& root_tsdf, uint32_t& root_weight) {
    constexpr uint32_t D = NodeLevels::MAX_DEPTH;
    uint8_t path[D] = {};
    uint32_t nodes_oct[D] = {};
    uint32_t nodes_tsdf[D][8] = {}, nodes_weight[D][8] = {};
    nodes_oct[0] = 0;  // the octree's root (octree.hpp:17-20)
    const float trunc_recip = 1.0f / trunc;  // :23
    uint32_t depth = 0;
    while (true) {
        const uint8_t child_i = path[depth]++;
        if (child_i >= 8) {  // :29-57
            const uint32_t a_tsdf = lv.nodes[depth].add(nodes_tsdf[depth]);
            const uint32_t a_weight = lv.nodes[depth].add(nodes_weight[depth]);
            std::memset(nodes_tsdf[depth], 0, 32); std::memset(nodes_weight[depth], 0, 32);
            if (depth == 0) { root_tsdf = a_tsdf; root_weight = a_weight; break; }
            depth--;
            const uint32_t at = path[depth] - 1u;
            nodes_tsdf[depth][at] = a_tsdf; nodes_weight[depth][at] = a_weight;
        } else if (depth < D - 1) {  // :59-68
            const uint32_t child = oct.nodes[nodes_oct[depth]].child[child_i];
            if (!child) continue;
            depth++; path[depth] = 0; nodes_oct[depth] = child;
        } else {  // :70-99
            const uint32_t child = oct.nodes[nodes_oct[depth]].child[child_i];
            if (!child) continue;
            const Node& node = oct.nodes[child];
            uint64_t c_tsdf = 0, c_weight = 0;
            for (uint32_t leaf_i = 0; leaf_i < 8; leaf_i++) {
                const uint32_t leaf = node.child[leaf_i];
                uint64_t sd_bits = 0xff, w_bits = 0xff;  // cluster.hpp:28-31,62-65
                if (leaf) {
                    sd_bits = cluster_sd_bits(oct.leaves[leaf].sd, trunc_recip);
                    if (saturate_weights) w_bits = std::min<uint32_t>(oct.leaves[leaf].w, 254u);
                }
                c_tsdf |= sd_bits << (leaf_i * 8); c_weight |= w_bits << (leaf_i * 8);
            }
            nodes_tsdf[depth][child_i] = lv.leaf_clusters.add(c_tsdf);
            nodes_weight[depth][child_i] = lv.leaf_clusters.add(c_weight);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The half of TSDFMap::save's mesh extraction that is the reference's own code (src/chad/detail/lvr2.cpp:15-133, ChadGrid's
// constructor): walk a submap's signed-distance tree, make a query point per leaf, put it into the eight cells
// around it, cull the cells with a missing corner.  What it hands to LVR2 (the triangulation, lvr2.cpp:235-270: third-party,
// absent, parity unpinned) is exactly this: the set of complete cells with their eight corner values.
inline uint32_t dag_child_addr(const NodeLevels& lv, uint32_t depth, uint32_t parent_addr, uint32_t child_i) {   // levels.hpp:147-161 (get_child_addr)
    const uint32_t child_mask = lv.nodes[depth].raw[parent_addr];
    const uint32_t child_bit = 1u << child_i;
    if (!(child_mask & child_bit)) return 0;
    const uint32_t child_count = uint32_t(__builtin_popcount(child_mask & (child_bit - 1u) & 0xffu));
    return lv.nodes[depth].raw[parent_addr + child_count + 1u];
}
// cluster.hpp:33-52 (TSDFs::try_get)
inline bool cluster_try_get(uint64_t value, uint32_t leaf_i, float sdf_trunc, float& signed_distance) {
    const uint64_t leaf_bits = (value >> uint64_t(leaf_i * 8)) & 0xff;
    if (leaf_bits == 0xff) return false;
    float sd = float(leaf_bits) - 127.0f;   // sd_range_abs = 255 / 2
    sd *= float(1.0 / float(127));
    sd *= sdf_trunc;
    signed_distance = sd;
    return true;
}
struct GridCell { uint64_t code; uint32_t vertex[8]; };   // MortonCode(cell_chunk)._value; query point per corner, 0xffffffff = FastBox::INVALID_INDEX
struct ChadGridOut {
    std::vector<V3> qp_pos; std::vector<float> qp_sd;      // m_queryPoints (lvr2.cpp:87-88)
    std::vector<GridCell> cells;                            // m_cells after the culling of lvr2.cpp:115-129, ascending by code (the reference's map is unordered)
};
void chad_grid(const NodeLevels& lv, uint32_t root_addr, float voxel_res, float trunc_dist, ChadGridOut& out) {
    constexpr uint32_t D = NodeLevels::MAX_DEPTH;
    uint8_t path_child[D] = {};
    uint32_t path_addr[D] = {};
    path_addr[0] = root_addr;                                // lvr2.cpp:24-29
    std::unordered_map<uint64_t, std::array<uint32_t, 8>> cells;
    uint32_t depth = 0;
    while (true) {                                           // :32-113
        const uint8_t child_i = path_child[depth]++;
        if (child_i == 8) {                                  // :37-40
            if (depth > 0) depth--; else break;
        } else if (depth < D - 1) {                          // :43-53
            const uint32_t child_addr = dag_child_addr(lv, depth, path_addr[depth], child_i);
            if (child_addr > 0) { depth++; path_child[depth] = 0; path_addr[depth] = child_addr; }
        } else {                                             // :55-112: leaf clusters
            const uint32_t cluster_addr = dag_child_addr(lv, D - 1, path_addr[depth], child_i);   // try_get_lc, levels.hpp:178-192
            if (!(lv.nodes[D - 1].raw[path_addr[depth]] & (1u << child_i))) continue;
            const uint64_t cluster = lv.leaf_clusters.raw[cluster_addr];
            uint64_t code = 0;                               // :61-66: the Morton code from the path
            for (uint64_t k = 0; k < 63 / 3 - 1; k++) code |= uint64_t(path_child[k] - 1) << uint64_t(60 - k * 3);
            int32_t cx, cy, cz;
            morton_decode(code, cx, cy, cz);                 // :67-68
            uint32_t leaf_i = 0;
            for (int32_t z = 0; z <= 1; z++) for (int32_t y = 0; y <= 1; y++) for (int32_t x = 0; x <= 1; x++, leaf_i++) {   // :70-73
                float signed_distance;
                if (!cluster_try_get(cluster, leaf_i, trunc_dist, signed_distance)) continue;   // :75-76
                const int32_t lx = cx + x, ly = cy + y, lz = cz + z;                              // :79
                const V3 leaf_pos{float(lx) * voxel_res, float(ly) * voxel_res, float(lz) * voxel_res};   // :80
                const uint32_t querypoint_i = uint32_t(out.qp_sd.size());                        // :88-89
                out.qp_pos.push_back(leaf_pos); out.qp_sd.push_back(signed_distance);
                static const int32_t cell_offsets[8][3] = {{0, 0, 0}, {-1, 0, 0}, {-1, -1, 0}, {0, -1, 0}, {0, 0, -1}, {-1, 0, -1}, {-1, -1, -1}, {0, -1, -1}};   // :92-102
                for (uint32_t i = 0; i < 8; i++) {           // :103-110
                    const uint64_t cell_code = morton_encode(lx + cell_offsets[i][0], ly + cell_offsets[i][1], lz + cell_offsets[i][2]);
                    auto [it, emplaced] = cells.try_emplace(cell_code);
                    if (emplaced) it->second.fill(0xffffffffu);
                    it->second[i] = querypoint_i;
                }
            }
        }
    }
    for (const auto& [code, vertex] : cells) {               // :115-129: cull incomplete cells
        bool incomplete = false;
        for (uint32_t i = 0; i < 8; i++) if (vertex[i] == 0xffffffffu) incomplete = true;
        if (incomplete) continue;
        GridCell c; c.code = code;
        for (uint32_t i = 0; i < 8; i++) c.vertex[i] = vertex[i];
        out.cells.push_back(c);
    }
    std::sort(out.cells.begin(), out.cells.end(), [](const GridCell& a, const GridCell& b) { return a.code < b.code; });
}

struct Snapshot {  // a closed submap's leaves, sorted by code (stands in for the out-of-scope DAG)
    std::vector<uint64_t> mc; std::vector<float> sd; std::vector<uint32_t> w;
};

struct Map {
    float res, trunc;
    Octree oct;
    std::vector<V3> positions;        // submap.hpp:110 (active submap)
    std::vector<Snapshot> closed;     // tsdf.cpp:54 (_submaps)
    NodeLevels levels;                // tsdf.cpp:30 (_node_levels_p): one DAG for all submaps
    std::vector<std::array<uint32_t, 2>> roots;  // submap.hpp:108-109 per closed submap
    bool saturate_weights = false;
    bool carve = false;               // EXTENSION (space carving, README.md:60 "not done" upstream): see carve_frame
    double ms_finalize = 0;
    // scratch reused across frames
    std::vector<PointMc> pmc; std::vector<V3> sorted, normals; std::vector<uint64_t> walked;
    // last-frame stats
    double ms_calc = 0, ms_sort = 0, ms_norm = 0, ms_upd = 0, ms_total = 0;
    uint64_t updates = 0, frames = 0;
};

void snapshot(const Octree& o, Snapshot& s) {
    const size_t n = o.leaves.size() - 1;
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = uint32_t(i + 1);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return o.leaf_mc[a] < o.leaf_mc[b]; });
    s.mc.resize(n); s.sd.resize(n); s.w.resize(n);
    for (size_t i = 0; i < n; i++) { s.mc[i] = o.leaf_mc[order[i]]; s.sd[i] = o.leaves[order[i]].sd; s.w[i] = o.leaves[order[i]].w; }
}

inline double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// octree.hpp:80-170
void octree_insert(Map& m, V3 pos) {
    const float res = m.res, trunc = m.trunc;
    const float res_recip = float(1.0 / double(res));  // :82
    uint64_t updates = 0;
    for (size_t i = 0; i < m.sorted.size(); i++) {
        const V3 p = m.sorted[i], nrm = m.normals[i];
        m.walked.clear();
        walk_ray(p, pos, res, res_recip, trunc, [&](I3 v) { m.walked.push_back(morton_encode(v.x, v.y, v.z)); });
        for (uint64_t mc : m.walked) {  // :153-164
            Leaf& leaf = m.oct.find_or_create(mc);
            int32_t vx, vy, vz;
            morton_decode(mc, vx, vy, vz);
            const V3 p2v{float(vx) * res - p.x, float(vy) * res - p.y, float(vz) * res - p.z};  // :157 (voxel corner)
            float d = dot(nrm, p2v);
            d = std::min(std::max(d, -trunc), trunc);  // std::clamp, :159
            leaf.sd = leaf.sd * float(leaf.w) + d;     // :161
            leaf.w++;
            leaf.sd = leaf.sd / float(leaf.w);         // :163
            m.oct.leaf_sum[size_t(&leaf - m.oct.leaves.data())] += double(d);
        }
        updates += m.walked.size();
    }
    m.updates = updates;
}

// EXTENSION, not in the reference (its roadmap lists space carving as open, README.md:59-60): after a frame's regular update, every
// ray also tells the voxels it crosses BEFORE its truncation band that they are free space.  Definition (the HIP library
// implements the same, CHAD_FLAG_SPACE_CARVING): for every usable point p, the traversal rule of octree.hpp:100-152 on the segment
// [position, p - trunc * dir]; every voxel on it other than the segment's last one (which is the first voxel of the regular walk)
// that ALREADY HOLDS A LEAF gets one observation of +trunc (sd = (sd * w + trunc) / (w + 1), w++); no leaf is created.  Points
// closer to the sensor than trunc are not carved.
void carve_frame(Map& m, V3 pos) {
    const float res = m.res, trunc = m.trunc, res_recip = float(1.0 / double(res));
    uint64_t carved = 0;
    for (size_t i = 0; i < m.sorted.size(); i++) {
        const V3 p = m.sorted[i];
        const V3 dir = normalize(p - pos);
        const V3 to = p - dir * trunc;
        if (!(dot(to - pos, dir) > 0.0f)) continue;
        const I3 last{int32_t(std::floor(to.x * res_recip)), int32_t(std::floor(to.y * res_recip)), int32_t(std::floor(to.z * res_recip))};
        walk_segment(pos, to, dir, res, res_recip, [&](I3 v) {
            if (v.x == last.x && v.y == last.y && v.z == last.z) return;
            Leaf* leaf = m.oct.find(morton_encode(v.x, v.y, v.z));
            if (!leaf || leaf->w == 0) return;
            leaf->sd = (leaf->sd * float(leaf->w) + trunc) / float(leaf->w + 1);
            leaf->w++;
            m.oct.leaf_sum[size_t(leaf - m.oct.leaves.data())] += double(trunc);
            carved++;
        });
    }
    m.updates += carved;
}

// tsdf.cpp:53-54 / :78-81: Submap::finalize into the shared DAG, then archive (the sorted leaf list is kept beside the
// DAG for the leaf-level parity tests)
void close_submap(Map& m) {
    auto t0 = std::chrono::steady_clock::now();
    uint32_t rt = 0, rw = 0;
    finalize(m.oct, m.levels, m.trunc, m.saturate_weights, rt, rw);
    m.ms_finalize = ms_since(t0);
    m.roots.push_back({rt, rw});
    m.closed.emplace_back();
    snapshot(m.oct, m.closed.back());
}

// tsdf.cpp:39-75
void map_insert(Map& m, const float* xyz, size_t n, const float position[3]) {
    auto t_total = std::chrono::steady_clock::now();
    const V3 pos{position[0], position[1], position[2]};
    if (m.positions.empty()) m.positions.push_back(pos);  // :48
    else {
        const V3 d = pos - m.positions.front();
        if (std::sqrt(dot(d, d)) > 5.0f) {  // :52
            close_submap(m);                   // :53-54
            m.positions.clear();
            m.positions.push_back(pos);        // :55-56
            m.oct.clear();                     // :57
        } else m.positions.push_back(pos);     // :60
    }
    auto t0 = std::chrono::steady_clock::now();
    calc_morton_vector(xyz, n, position, m.res, m.pmc);  // :64
    m.ms_calc = ms_since(t0); t0 = std::chrono::steady_clock::now();
    sort_morton_vector(m.pmc, m.sorted);                 // :65
    m.ms_sort = ms_since(t0); t0 = std::chrono::steady_clock::now();
    estimate_normals(m.pmc, pos, m.normals, nullptr);    // :67
    m.ms_norm = ms_since(t0); t0 = std::chrono::steady_clock::now();
    octree_insert(m, pos);                               // :70
    if (m.carve) carve_frame(m, pos);                    // (extension, off by default)
    m.ms_upd = ms_since(t0);
    m.ms_total = ms_since(t_total);
    m.frames++;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C entry points (loaded with ctypes by tests/ and bench.py's cpu_baseline leg)
extern "C" {

void* oracle_map_create(float res, float trunc) { Map* m = new Map(); m->res = res; m->trunc = trunc; return m; }
void oracle_map_destroy(void* h) { delete static_cast<Map*>(h); }
void oracle_map_insert(void* h, const float* xyz, size_t n, const float* pos3) { map_insert(*static_cast<Map*>(h), xyz, n, pos3); }
// empties the active submap without archiving it (what a fresh TSDFMap would hold)
void oracle_map_reset(void* h) { Map* m = static_cast<Map*>(h); m->oct.clear(); m->positions.clear(); m->closed.clear(); m->levels = NodeLevels(); m->roots.clear(); }
size_t oracle_map_closed_count(void* h) { return static_cast<Map*>(h)->closed.size(); }

// submap < 0: the active octree; otherwise the submap closed at that roll-over.  Leaves sorted by code, ascending.
size_t oracle_map_leaf_count(void* h, int submap) {
    Map* m = static_cast<Map*>(h);
    if (submap < 0) return m->oct.leaves.size() - 1;
    return size_t(submap) < m->closed.size() ? m->closed[size_t(submap)].mc.size() : 0;
}
size_t oracle_map_export_leaves(void* h, int submap, uint64_t* mc, float* sd, uint32_t* w, size_t cap) {
    Map* m = static_cast<Map*>(h);
    Snapshot tmp;
    const Snapshot* s;
    if (submap < 0) { snapshot(m->oct, tmp); s = &tmp; }
    else if (size_t(submap) < m->closed.size()) s = &m->closed[size_t(submap)];
    else return 0;
    const size_t n = std::min(cap, s->mc.size());
    if (n) { std::memcpy(mc, s->mc.data(), n * 8); std::memcpy(sd, s->sd.data(), n * 4); std::memcpy(w, s->w.data(), n * 4); }   // (memcpy's pointers must not be null, even for 0 bytes: UBSan)
    return s->mc.size();
}
// The exact means (f64 sum / count) of the ACTIVE octree's leaves, in the order of oracle_map_export_leaves(h, -1, ...).
size_t oracle_map_export_exact_means(void* h, double* mean, size_t cap) {
    Map* m = static_cast<Map*>(h);
    const size_t n = m->oct.leaves.size() - 1;
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = uint32_t(i + 1);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return m->oct.leaf_mc[a] < m->oct.leaf_mc[b]; });
    for (size_t i = 0; i < std::min(n, cap); i++) mean[i] = m->oct.leaf_sum[order[i]] / double(m->oct.leaves[order[i]].w);
    return n;
}
// out[0..9]: ms calc, sort, normals, update, total; updates U, leaves V, hash entries B, octree nodes, frames
void oracle_map_stats(void* h, double* out) {
    Map* m = static_cast<Map*>(h);
    out[0] = m->ms_calc; out[1] = m->ms_sort; out[2] = m->ms_norm; out[3] = m->ms_upd; out[4] = m->ms_total;
    out[5] = double(m->updates); out[6] = double(m->oct.leaves.size() - 1); out[7] = double(m->oct.lookup.count);
    out[8] = double(m->oct.nodes.size()); out[9] = double(m->frames);
}

// ---- Submap::finalize / NodeLevels (submap.hpp, levels.hpp) ----
void oracle_map_set_saturate_weights(void* h, int on) { static_cast<Map*>(h)->saturate_weights = on != 0; }
void oracle_map_set_carving(void* h, int on) { static_cast<Map*>(h)->carve = on != 0; }
// TSDFMap::save's hand-over (tsdf.cpp:78-81): finalize and archive the active submap if it holds a pose
void oracle_map_close_submap(void* h) {
    Map* m = static_cast<Map*>(h);
    if (m->positions.empty()) return;
    close_submap(*m);
    m->positions.clear();
    m->oct.clear();
}
// level 0..19: node levels (u32 words), 20: the leaf clusters (u64 words); counts include the reserved word 0
size_t oracle_dag_level_words(void* h, uint32_t level) {
    Map* m = static_cast<Map*>(h);
    return level < 20 ? m->levels.nodes[level].raw.size() : level == 20 ? m->levels.leaf_clusters.raw.size() : 0;
}
void oracle_dag_level_copy(void* h, uint32_t level, void* out) {
    Map* m = static_cast<Map*>(h);
    if (level < 20) std::memcpy(out, m->levels.nodes[level].raw.data(), m->levels.nodes[level].raw.size() * 4);
    else if (level == 20) std::memcpy(out, m->levels.leaf_clusters.raw.data(), m->levels.leaf_clusters.raw.size() * 8);
}
// out[0] = uniques, out[1] = dupes (levels.hpp:88,136)
void oracle_dag_level_counts(void* h, uint32_t level, uint32_t* out) {
    Map* m = static_cast<Map*>(h);
    if (level < 20) { out[0] = m->levels.nodes[level].uniques; out[1] = m->levels.nodes[level].dupes; }
    else { out[0] = m->levels.leaf_clusters.uniques; out[1] = m->levels.leaf_clusters.dupes; }
}
int oracle_dag_roots(void* h, uint32_t submap, uint32_t* out) {
    Map* m = static_cast<Map*>(h);
    if (submap >= m->roots.size()) return 1;
    out[0] = m->roots[submap][0]; out[1] = m->roots[submap][1];
    return 0;
}
double oracle_map_finalize_ms(void* h) { return static_cast<Map*>(h)->ms_finalize; }
// Finalize a submap given as a leaf list (any order): the leaves are put into a fresh octree with the given values and
// that octree is folded into this map's DAG as its next closed submap.  Lets a test feed the device's own (sd, w).
void oracle_map_close_submap_from_leaves(void* h, const uint64_t* mc, const float* sd, const uint32_t* w, size_t n) {
    Map* m = static_cast<Map*>(h);
    Octree o;
    for (size_t i = 0; i < n; i++) { Leaf& l = o.find_or_create(mc[i]); l.sd = sd[i]; l.w = w[i]; }
    uint32_t rt = 0, rw = 0;
    auto t0 = std::chrono::steady_clock::now();
    finalize(o, m->levels, m->trunc, m->saturate_weights, rt, rw);
    m->ms_finalize = ms_since(t0);
    m->roots.push_back({rt, rw});
    m->closed.emplace_back();
    snapshot(o, m->closed.back());
}

// ChadGrid (lvr2.cpp:15-133) of closed submap `submap`: the complete cells, ascending by the Morton code of their lowest corner;
// corners[8 * c + i] = signed distance of the query point at vertex i of cell c (vertex order of lvr2.cpp:92-102).  Returns the
// number of complete cells; writes at most cap of them.  query_points (optional): out[0] = number of query points.
size_t oracle_chad_grid(void* h, uint32_t submap, uint64_t* codes, float* corners, size_t cap, uint64_t* query_points) {
    Map* m = static_cast<Map*>(h);
    if (submap >= m->roots.size()) return 0;
    ChadGridOut g;
    chad_grid(m->levels, m->roots[submap][0], m->res, m->trunc, g);
    if (query_points) *query_points = g.qp_sd.size();
    for (size_t c = 0; c < g.cells.size() && c < cap; c++) {
        if (codes) codes[c] = g.cells[c].code;
        if (corners) for (uint32_t i = 0; i < 8; i++) corners[8 * c + i] = g.qp_sd[g.cells[c].vertex[i]];
    }
    return g.cells.size();
}

// Marching cubes over ChadGrid's complete cells -- what LVR2's FastReconstruction does with them (lvr2.cpp:246-251: BoxT::getSurface per
// cell).  LVR2 is a third-party dependency absent from this image (cmake/lvr2.cmake:10, release 25.1.0), so this restates the PUBLISHED
// algorithm its FastBox family implements, not LVR2's code: the case index has bit i set where corner i's distance is positive
// (FastBox::getIndex), the 256-case table of Lorensen & Cline as tabulated by Bourke (marching_cubes_table.inc, the oracle's own copy,
// generated and checked by scripts/mc_table_check.py) lists up to five triangles by the cell edges their corners lie on, and a corner
// is placed by linear interpolation between its edge's two query points.  BilinearFastBox::optimizePlanarFaces (lvr2.cpp:262) is not
// modelled.  Positions in f32 with the interpolation written from the edge's first corner in (z, y, x) order, as the device's
// kernel writes it.  triangles: 9 floats each, in cell order then table order; returns their number, writes at most cap.
#include "marching_cubes_table.inc"
size_t oracle_marching_cubes(const uint64_t* codes, const float* corners, size_t cells, float res, float* triangles, size_t cap) {
    static const int ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};   // lvr2.cpp:92-102
    static const int edge_a[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, edge_b[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
    size_t count = 0;
    for (size_t c = 0; c < cells; c++) {
        const float* val = corners + 8 * c;
        int32_t bx, by, bz;
        morton_decode(codes[c], bx, by, bz);
        uint32_t index = 0;
        for (int v = 0; v < 8; v++) if (val[v] > 0.0f) index |= 1u << v;
        for (int t = 0; kOracleMarchingCubes[index][t] >= 0; t += 3, count++) {
            if (count >= cap) continue;
            for (int k = 0; k < 3; k++) {
                const int e = kOracleMarchingCubes[index][t + k];
                int p = edge_a[e], q = edge_b[e];
                if (oz[p] * 4 + oy[p] * 2 + ox[p] > oz[q] * 4 + oy[q] * 2 + ox[q]) std::swap(p, q);
                const float sp = val[p], sq = val[q];
                const float w = sp / (sp - sq);
                const float px = float(bx + ox[p]) * res, py = float(by + oy[p]) * res, pz = float(bz + oz[p]) * res;
                const float qx = float(bx + ox[q]) * res, qy = float(by + oy[q]) * res, qz = float(bz + oz[q]) * res;
                float* out = triangles + 9 * count + 3 * k;
                out[0] = px + (qx - px) * w; out[1] = py + (qy - py) * w; out[2] = pz + (qz - pz) * w;
            }
        }
    }
    return count;
}

// ---- stage-level entry points for unit parity tests ----
// morton.hpp:21-28,59-80: codes of n points (no D2 filtering: callers pass valid points)
void oracle_morton_codes(const float* xyz, size_t n, float res, uint64_t* out) {
    const float recip = float(1.0 / double(res));
    for (size_t i = 0; i < n; i++)
        out[i] = morton_encode(int32_t(std::floor(xyz[3 * i] * recip)), int32_t(std::floor(xyz[3 * i + 1] * recip)), int32_t(std::floor(xyz[3 * i + 2] * recip)));
}
void oracle_morton_encode_ivec(const int32_t* v, size_t n, uint64_t* out) { for (size_t i = 0; i < n; i++) out[i] = morton_encode(v[3 * i], v[3 * i + 1], v[3 * i + 2]); }
void oracle_morton_decode_ivec(const uint64_t* mc, size_t n, int32_t* v) { for (size_t i = 0; i < n; i++) morton_decode(mc[i], v[3 * i], v[3 * i + 1], v[3 * i + 2]); }
// portable (non-BMI2) interleave, to cross-check the pdep/pext path against the published bit layout
uint64_t oracle_morton_encode_portable(int32_t vx, int32_t vy, int32_t vz) {
    uint32_t x = (1u << 20) + uint32_t(vx), y = (1u << 20) + uint32_t(vy), z = (1u << 20) + uint32_t(vz);
    return spread21(x) | (spread21(y) << 1) | (spread21(z) << 2);
}

// Stages a4+a5+a6 on one frame: writes the sorted order (input indices), sorted codes, normals per sorted
// index and the group-start flags.  Returns the number of points kept (D2).
size_t oracle_frame_normals(const float* xyz, size_t n, const float* pos3, float res,
                            uint32_t* order, uint64_t* mc_sorted, float* normals, uint8_t* group_start) {
    std::vector<PointMc> pmc; std::vector<V3> sorted, nrm;
    calc_morton_vector(xyz, n, pos3, res, pmc);
    sort_morton_vector(pmc, sorted);
    std::vector<uint8_t> gs(pmc.size(), 0);
    estimate_normals(pmc, V3{pos3[0], pos3[1], pos3[2]}, nrm, gs.data());
    for (size_t i = 0; i < pmc.size(); i++) {
        if (order) order[i] = pmc[i].idx;
        if (mc_sorted) mc_sorted[i] = pmc[i].mc;
        if (normals) { normals[3 * i] = nrm[i].x; normals[3 * i + 1] = nrm[i].y; normals[3 * i + 2] = nrm[i].z; }
        if (group_start) group_start[i] = gs[i];
    }
    return pmc.size();
}

// a8: voxel list of one ray; returns the count (writes at most cap voxels)
size_t oracle_walk_ray(const float* p3, const float* pos3, float res, float trunc, int32_t* voxels, size_t cap) {
    size_t k = 0;
    walk_ray(V3{p3[0], p3[1], p3[2]}, V3{pos3[0], pos3[1], pos3[2]}, res, float(1.0 / double(res)), trunc, [&](I3 v) {
        if (k < cap) { voxels[3 * k] = v.x; voxels[3 * k + 1] = v.y; voxels[3 * k + 2] = v.z; }
        k++;
    });
    return k;
}
// a8 for a batch: counts[i] voxels of ray i written at voxels[(i*stride + j)*3]
void oracle_walk_rays(const float* xyz, size_t n, const float* pos3, float res, float trunc, int32_t* voxels, uint32_t* counts, size_t stride) {
    for (size_t i = 0; i < n; i++) counts[i] = uint32_t(oracle_walk_ray(xyz + 3 * i, pos3, res, trunc, voxels + 3 * i * stride, stride));
}

}  // extern "C"

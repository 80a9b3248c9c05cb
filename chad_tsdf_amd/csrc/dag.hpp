// dag.hpp -- host half of Submap::finalize (submap.hpp:10-106): the map-wide DAG of NodeLevels (levels.hpp:141-199).
//
// The reference walks its pointer octree depth-first in child order and adds every node child-first to a per-level
// store that hands out addresses in order of first occurrence (levels.hpp:56-86, :122-134), so the DAG is a function of
// the leaf clusters in ascending Morton order and of nothing else.  The device delivers exactly that stream
// (k_leaf_clusters: eight 64-bit cluster words per 4x4x4 block, blocks sorted by key); this file folds it into the
// levels with a stack of the twenty nodes on the current root-to-block path.  Address-for-address the same arrays as
// the reference's NodeLevel::_raw_data / LeafClusterLevel::_raw_data.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace chadhip {

constexpr uint32_t kDagDepth = 20;               // NodeLevels::MAX_DEPTH (levels.hpp:193)
constexpr uint64_t kClusterEmpty = ~uint64_t(0); // eight 0xff bytes: no leaf in the octet (cluster.hpp:28-31)

inline uint64_t dag_mix(uint64_t h) { h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29; return h; }

// Open-addressed set of addresses into a level's raw words; 0 = free (address 0 is reserved by the reference, levels.hpp:52).
struct AddrSet {
    std::vector<uint32_t> slot = std::vector<uint32_t>(1024, 0u);
    uint32_t used = 0;
    template <class Hash>
    void grow(Hash&& hash_of) {
        std::vector<uint32_t> old(slot.size() * 2, 0u);
        old.swap(slot);
        const size_t mask = slot.size() - 1;
        for (uint32_t a : old) {
            if (!a) continue;
            size_t i = size_t(hash_of(a)) & mask;
            while (slot[i]) i = (i + 1) & mask;
            slot[i] = a;
        }
    }
};

struct DagNodeLevel {  // NodeLevel, levels.hpp:8-94: [child mask, child addresses of the set bits...] per node
    std::vector<uint32_t> raw{0u};
    uint32_t uniques = 0, dupes = 0;
    AddrSet set;
    uint64_t hash_at(uint32_t addr) const {
        const uint32_t n = uint32_t(__builtin_popcount(raw[addr] & 0xffu));
        uint64_t h = raw[addr];
        for (uint32_t i = 1; i <= n; i++) h = dag_mix(h ^ raw[addr + i]) + i;
        return dag_mix(h);
    }
    uint32_t add(const uint32_t children[8]) {  // levels.hpp:56-86
        const uint32_t addr = uint32_t(raw.size());
        raw.push_back(0u);
        uint32_t n = 0;
        for (uint32_t i = 0; i < 8; i++) {
            if (!children[i]) continue;
            raw.push_back(children[i]);
            raw[addr] |= 1u << i;
            n++;
        }
        if ((set.used + 1) * 2 > set.slot.size()) set.grow([this](uint32_t a) { return hash_at(a); });
        const size_t mask = set.slot.size() - 1;
        for (size_t i = size_t(hash_at(addr)) & mask;; i = (i + 1) & mask) {
            const uint32_t other = set.slot[i];
            if (!other) { set.slot[i] = addr; set.used++; uniques++; return addr; }
            if (raw[other] == raw[addr] && std::memcmp(&raw[other + 1], &raw[addr + 1], n * 4) == 0) {
                raw.resize(addr);
                dupes++;
                return other;
            }
        }
    }
};

struct DagClusterLevel {  // LeafClusterLevel, levels.hpp:95-139
    std::vector<uint64_t> raw{0ull};
    uint32_t uniques = 0, dupes = 0;
    // the look-up table keeps the value beside its address: one cache line per probe, and the line can be asked for ahead of time
    struct Slot { uint64_t value; uint64_t addr; };
    std::vector<Slot> slot = std::vector<Slot>(1024, Slot{0, 0});
    uint32_t used = 0;
    void prefetch(uint64_t value) const { __builtin_prefetch(&slot[size_t(dag_mix(value)) & (slot.size() - 1)]); }
    void grow() {
        std::vector<Slot> old(slot.size() * 2, Slot{0, 0});
        old.swap(slot);
        const size_t mask = slot.size() - 1;
        for (const Slot& e : old) {
            if (!e.addr) continue;
            size_t i = size_t(dag_mix(e.value)) & mask;
            while (slot[i].addr) i = (i + 1) & mask;
            slot[i] = e;
        }
    }
    uint32_t add(uint64_t value) {  // levels.hpp:122-134
        if ((used + 1) * 2 > slot.size()) grow();
        const size_t mask = slot.size() - 1;
        for (size_t i = size_t(dag_mix(value)) & mask;; i = (i + 1) & mask) {
            Slot& e = slot[i];
            if (!e.addr) {
                const uint32_t addr = uint32_t(raw.size());
                raw.push_back(value);
                e.value = value; e.addr = addr; used++; uniques++;
                return addr;
            }
            if (e.value == value) { dupes++; return uint32_t(e.addr); }
        }
    }
};

struct Dag {
    DagNodeLevel nodes[kDagDepth];
    DagClusterLevel clusters;
};

// Child index taken at depth d on the way to Morton code mc (octree.hpp:45-47: three bits per level, root first).
inline uint32_t dag_digit(uint64_t mc, uint32_t depth) { return uint32_t(mc >> (60u - 3u * depth)) & 7u; }

// Fold one submap into the DAG.  keys: block keys (MortonCode & ~63) ascending; tsdf[8 * b + o] / weight[8 * b + o]: the
// cluster words of octet o of block b, kClusterEmpty where the octet holds no leaf (weight == nullptr: every present
// octet's weight word is all ones, which is what the reference stores -- Q6, submap.hpp:93).
inline void dag_add_submap(Dag& dag, const uint64_t* keys, const uint64_t* tsdf, const uint64_t* weight, size_t blocks,
                           uint32_t* root_tsdf, uint32_t* root_weight) {
    uint32_t kids_t[kDagDepth][8] = {}, kids_w[kDagDepth][8] = {};
    uint32_t top_t = 0, top_w = 0;
    // close the finished nodes on the path to `code`, deepest first, down to depth `last` (submap.hpp:29-57)
    auto close = [&](uint64_t code, uint32_t last) {
        for (uint32_t d = kDagDepth; d-- > last;) {
            const uint32_t a_t = dag.nodes[d].add(kids_t[d]);
            const uint32_t a_w = dag.nodes[d].add(kids_w[d]);
            std::memset(kids_t[d], 0, 32); std::memset(kids_w[d], 0, 32);
            if (d == 0) { top_t = a_t; top_w = a_w; break; }
            const uint32_t at = dag_digit(code, d - 1);
            kids_t[d - 1][at] = a_t; kids_w[d - 1][at] = a_w;
        }
    };
    bool any = false;
    uint64_t last_key = 0;
    uint32_t all_ones = 0;  // address of the all-ones weight cluster once it has been looked up
    for (size_t b = 0; b < blocks; b++) {
        bool holds_leaves = false;
        for (uint32_t o = 0; o < 8; o++) holds_leaves |= tsdf[8 * b + o] != kClusterEmpty;
        if (!holds_leaves) continue;  // a brick that was claimed but never written has no counterpart in the octree
        if (any) {  // nodes below the first level at which the two paths part are complete
            const uint64_t x = last_key ^ keys[b];
            const uint32_t parted = (62u - (63u - uint32_t(__builtin_clzll(x)))) / 3u;
            close(last_key, parted + 1);
        }
        any = true; last_key = keys[b];
        if (b + 2 < blocks) for (uint32_t o = 0; o < 8; o++) dag.clusters.prefetch(tsdf[8 * (b + 2) + o]);
        for (uint32_t o = 0; o < 8; o++) {  // submap.hpp:70-99: signed distances first, then weights
            const uint64_t t = tsdf[8 * b + o];
            if (t == kClusterEmpty) continue;
            kids_t[kDagDepth - 1][o] = dag.clusters.add(t);
            if (weight) kids_w[kDagDepth - 1][o] = dag.clusters.add(weight[8 * b + o]);
            else if (all_ones) { kids_w[kDagDepth - 1][o] = all_ones; dag.clusters.dupes++; }   // the one weight word there is: no look-up
            else kids_w[kDagDepth - 1][o] = all_ones = dag.clusters.add(kClusterEmpty);
        }
    }
    if (any) close(last_key, 0);
    else {  // an empty octree still yields a childless root pair (submap.hpp:29-44 at depth 0)
        top_t = dag.nodes[0].add(kids_t[0]);
        top_w = dag.nodes[0].add(kids_w[0]);
    }
    *root_tsdf = top_t; *root_weight = top_w;
}

// Depth-first read-back of one submap's signed-distance tree (the traversal of lvr2.cpp:30-90): calls
// leaf(mc, byte) for every stored leaf, ascending by Morton code; byte is the 8-bit signed distance (cluster.hpp:33-52).
template <class Leaf>
inline void dag_walk(const Dag& dag, uint32_t root, Leaf&& leaf) {
    uint8_t child[kDagDepth] = {};
    uint32_t addr[kDagDepth] = {};
    addr[0] = root;
    uint32_t depth = 0;
    while (true) {
        const uint32_t c = child[depth]++;
        if (c == 8) {
            if (depth == 0) break;
            depth--;
            continue;
        }
        const std::vector<uint32_t>& raw = dag.nodes[depth].raw;
        const uint32_t mask = raw[addr[depth]];
        if (!(mask & (1u << c))) continue;
        const uint32_t at = raw[addr[depth] + 1 + uint32_t(__builtin_popcount(mask & ((1u << c) - 1u)))];  // levels.hpp:142-156
        if (depth < kDagDepth - 1) {
            depth++; child[depth] = 0; addr[depth] = at;
            continue;
        }
        uint64_t code = 0;
        for (uint32_t k = 0; k < kDagDepth; k++) code |= uint64_t(child[k] - 1u) << (60u - 3u * k);  // lvr2.cpp:62-67
        const uint64_t word = dag.clusters.raw[at];
        for (uint32_t i = 0; i < 8; i++) {
            const uint32_t bits = uint32_t(word >> (8 * i)) & 0xffu;
            if (bits != 0xffu) leaf(code | i, uint8_t(bits));
        }
    }
}

// A position in one submap's DAG: the reference's (commented-out) TSDFMap::iterator -- tsdf.hpp:125-151: _node_paths, _node_addrs,
// _leaf_cluster_p, _mc_raw; begin() at tsdf.cpp:104-146 walks to the first leaf exactly like first() below -- completed with the
// operations the reference leaves as TODO (operator++ / -- / + / -, end()).  Leaves are visited in ascending Morton order.  The
// signed-distance tree and the weight tree of a submap have the same shape (submap.hpp:70-99 adds them side by side), so one path
// serves both.
struct DagCursor {
    const Dag* dag = nullptr;
    uint32_t addr_t[kDagDepth] = {}, addr_w[kDagDepth] = {};   // node addresses on the path, per tree
    int8_t child[kDagDepth] = {};                              // child taken at each depth
    uint32_t cluster_t = 0, cluster_w = 0;                     // addresses of the current leaf clusters
    int32_t leaf = 0;                                          // leaf inside the cluster, 0..7
    bool valid = false;

    void open(const Dag& d, uint32_t root_tsdf, uint32_t root_weight) {
        dag = &d; addr_t[0] = root_tsdf; addr_w[0] = root_weight; valid = false;
    }
    uint64_t code() const {   // tsdf.cpp:138-143 / lvr2.cpp:62-67: three bits per level, root first; the leaf's bits last
        uint64_t c = uint64_t(leaf);
        for (uint32_t k = 0; k < kDagDepth; k++) c |= uint64_t(child[k]) << (60u - 3u * k);
        return c;
    }
    uint8_t sd_bits() const { return uint8_t(dag->clusters.raw[cluster_t] >> (8 * leaf)); }
    uint8_t weight_bits() const { return uint8_t(dag->clusters.raw[cluster_w] >> (8 * leaf)); }

    bool first() { return valid = descend(0, +1) && settle(+1); }
    bool last() { return valid = descend(0, -1) && settle(-1); }
    bool next() { return step(+1); }
    bool prev() { return step(-1); }

private:
    uint32_t mask_at(uint32_t depth) const { return dag->nodes[depth].raw[addr_t[depth]] & 0xffu; }
    uint32_t child_addr(const uint32_t* addr, uint32_t depth, uint32_t c) const {   // levels.hpp:142-156
        const std::vector<uint32_t>& raw = dag->nodes[depth].raw;
        const uint32_t mask = raw[addr[depth]];
        return raw[addr[depth] + 1 + uint32_t(__builtin_popcount(mask & ((1u << c) - 1u)))];
    }
    // lowest (dir > 0) / highest (dir < 0) set bit of `mask` strictly beyond position `from` (-1 .. 8) in direction dir; -1 if none
    static int pick(uint32_t mask, int from, int dir) {
        mask &= 0xffu;
        if (dir > 0) { const uint32_t m = from < 0 ? mask : (from >= 7 ? 0u : mask & (0xffu << (from + 1))); return m ? __builtin_ctz(m) : -1; }
        const uint32_t m = from > 7 ? mask : (from <= 0 ? 0u : mask & ((1u << from) - 1u));
        return m ? 31 - __builtin_clz(m) : -1;
    }
    // from the node at `depth` (its address set) down to a leaf cluster, always taking the first child in direction dir
    bool descend(uint32_t depth, int dir) {
        for (uint32_t d = depth; d < kDagDepth; d++) {
            const int c = pick(mask_at(d), dir > 0 ? -1 : 8, dir);
            if (c < 0) return false;   // only an empty submap's childless root (submap.hpp:29-44)
            child[d] = int8_t(c);
            const uint32_t at = child_addr(addr_t, d, uint32_t(c)), aw = child_addr(addr_w, d, uint32_t(c));
            if (d + 1 < kDagDepth) { addr_t[d + 1] = at; addr_w[d + 1] = aw; }
            else { cluster_t = at; cluster_w = aw; }
        }
        leaf = dir > 0 ? -1 : 8;
        return true;
    }
    // move `leaf` to the next stored leaf of the current cluster in direction dir (0xff = no leaf, cluster.hpp:28-31)
    bool settle(int dir) {
        const uint64_t word = dag->clusters.raw[cluster_t];
        for (int i = leaf + dir; i >= 0 && i < 8; i += dir)
            if ((uint32_t(word >> (8 * i)) & 0xffu) != 0xffu) { leaf = i; return true; }
        return false;
    }
    bool step(int dir) {
        if (!valid) return false;
        if (settle(dir)) return true;
        // leave the cluster: climb to the deepest node with a further child in that direction, descend again from there
        for (;;) {
            int d = int(kDagDepth) - 1, c = -1;
            for (; d >= 0; d--) { c = pick(mask_at(uint32_t(d)), child[d], dir); if (c >= 0) break; }
            if (d < 0) return valid = false;   // past the last (first) leaf
            child[d] = int8_t(c);
            const uint32_t at = child_addr(addr_t, uint32_t(d), uint32_t(c)), aw = child_addr(addr_w, uint32_t(d), uint32_t(c));
            if (uint32_t(d) + 1 < kDagDepth) {
                addr_t[d + 1] = at; addr_w[d + 1] = aw;
                if (!descend(uint32_t(d) + 1, dir)) return valid = false;   // (a stored node always has a child)
            } else { cluster_t = at; cluster_w = aw; leaf = dir > 0 ? -1 : 8; }
            if (settle(dir)) return true;
            // (a stored cluster always holds a leaf -- dag_add_submap skips empty octets; if not, simply go on from here)
        }
    }
};

// one axis of MortonCode::decode (morton.hpp:29-37; libmorton's morton3D_64_decode): every third bit, compacted
inline uint32_t dag_compact21(uint64_t v) {
    v &= 0x1249249249249249ull;
    v = (v | (v >> 2)) & 0x10c30c30c30c30c3ull;
    v = (v | (v >> 4)) & 0x100f00f00f00f00full;
    v = (v | (v >> 8)) & 0x001f0000ff0000ffull;
    v = (v | (v >> 16)) & 0x001f00000000ffffull;
    v = (v | (v >> 32)) & 0x1fffffull;
    return uint32_t(v);
}

// cluster.hpp:33-52
inline float cluster_sd_of_bits(uint8_t bits, float trunc) {
    float sd = float(bits) - 127.0f;
    sd *= float(1.0 / 127.0f);
    return sd * trunc;
}

}  // namespace chadhip

// query_kernels.hpp -- read access to the ACTIVE submap on the device: the query half of the reference's intended read-back API
// (tsdf.hpp:120-161 lists it as TODO: "leaf iterator / raycast to retrieve leaves along it + physics hit").  A leaf is the
// reference's Leaf{_signed_distance, _weight} (octree.hpp:15) of the voxel whose CORNER sits at decode(mc) * res -- the point the
// update measures its distance from (octree.hpp:157) -- and reads back as sum / count exactly as chad_map_export_leaves does.
//   k_query_points   per query point: the leaf of the voxel containing it, or the trilinear interpolation of the eight leaves
//                    around it (the corners of the cell the point lies in -- the cell of the reference's mesh extraction,
//                    lvr2.cpp:93-113); NaN / weight 0 where a leaf is missing
//   k_raycast        per ray: march in steps of res / 2, sample the trilinear field, first crossing from positive (free space in
//                    front of the surface) to non-positive; the hit distance by linear interpolation between the two samples
// Both only read the map (plain loads); run them on the map's stream behind the frames (the host entry points do).
#pragma once
#include "map_kernels.hpp"

namespace chadhip {

// slot of block `key` in the active table, or kNoBrick (read-only probe: same sequence as find_or_create_brick)
__device__ __forceinline__ uint32_t find_brick(const MapView& m, uint64_t key) {
    uint32_t slot = hash_block(key, m.hash_mask);
    for (uint32_t probes = 0; probes < kMaxProbes; probes++) {
        const uint64_t k = m.hash_keys[slot];
        if (k == key) return slot;
        if (k == kEmptyKey) return kNoBrick;
        slot = (slot + 1) & m.hash_mask;
    }
    return kNoBrick;
}

// Leaf of voxel v: false if it was never observed.  sd = (integer sum -> f32, one rounding) / count, as the export computes it.
__device__ __forceinline__ bool read_leaf(const MapView& m, I3 v, double metres_per_unit, float* sd, uint32_t* w) {
    const uint64_t mc = morton_encode(v.x, v.y, v.z);
    const uint32_t slot = find_brick(m, mc & kBlockMask);
    if (slot == kNoBrick) return false;
    const size_t at = size_t(slot) * 64 + (uint32_t(mc) & 63u);
    long long units; uint32_t count;
    voxel_totals(m, at, &units, &count);
    if (!(count & ~kPoisonedCount)) return false;
    const float sum = float(double(units) * metres_per_unit);
    *sd = (count & kPoisonedCount) ? __int_as_float(0x7fc00000) : sum / float(count & ~kPoisonedCount);
    *w = count & ~kPoisonedCount;
    return true;
}

// Trilinear sample at p: the eight leaves at floor(p / res) + {0,1}^3, weights by the position inside the cell (f32, fixed
// operation order: x first, then y, then z).  weight = the smallest of the eight; false if any is missing.
__device__ __forceinline__ bool sample_trilinear(const MapView& m, F3 p, float res, float recip, double metres_per_unit, float* sd, uint32_t* w) {
    const I3 v = voxel_of(p, recip);
    const float fx = (p.x - float(v.x) * res) * recip, fy = (p.y - float(v.y) * res) * recip, fz = (p.z - float(v.z) * res) * recip;
    float c[8];
    uint32_t wmin = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t wk;
        if (!read_leaf(m, I3{v.x + (k & 1), v.y + ((k >> 1) & 1), v.z + (k >> 2)}, metres_per_unit, &c[k], &wk)) return false;
        wmin = min(wmin, wk);
    }
    const float c00 = c[0] + (c[1] - c[0]) * fx, c10 = c[2] + (c[3] - c[2]) * fx, c01 = c[4] + (c[5] - c[4]) * fx, c11 = c[6] + (c[7] - c[6]) * fx;
    const float c0 = c00 + (c10 - c00) * fy, c1 = c01 + (c11 - c01) * fy;
    *sd = c0 + (c1 - c0) * fz;
    *w = wmin;
    return true;
}

// ---- space carving: an EXTENSION (upstream's roadmap lists it as open, README.md:59-60) -------------------------
// After a frame's regular update every ray also tells the voxels it crosses before its truncation band that they are free space:
// the traversal rule of octree.hpp:100-152 on the segment [position, p - trunc * dir]; every voxel on it other than the segment's
// last one (the first voxel of the regular walk) that ALREADY HOLDS A LEAF gets one observation of +trunc; no leaf, no block is
// created.  Points closer to the sensor than trunc are not carved.  (The test oracle restates the same definition: carve_frame.)
// Runs behind the frame's update on the same stream, frames one after the other (the set of leaves must not move under it).
__global__ void __launch_bounds__(256)
k_carve(const float4* __restrict__ points0, const float4* __restrict__ points1, FrameHeader* __restrict__ h, F3 pos, float res, float recip, float trunc, MapView m) {
    const float4* __restrict__ pts = (h->sort_passes & 1u) ? points0 : points1;   // the frame in processing order (any order would do)
    const uint32_t n = h->n_usable;
    unsigned long long carved = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) mark_exact_rows(m);   // carving adds to the exact rows (map_kernels.hpp, kTableExactRows)
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const float4 q = pts[i];
        const F3 p{q.x, q.y, q.z};
        const F3 dir = unit3(sub3(p, pos));                                          // octree.hpp:92
        const F3 to{p.x - dir.x * trunc, p.y - dir.y * trunc, p.z - dir.z * trunc};  // :94: where the regular walk begins
        if (!(dot3(sub3(to, pos), dir) > 0.0f)) continue;
        RayWalk w;
        w.begin_segment(pos, to, dir, res, recip);
        uint64_t cached_key = kEmptyKey;
        uint32_t brick = kNoBrick;
        do {
            if (w.cur.x == w.last.x && w.cur.y == w.last.y && w.cur.z == w.last.z) continue;
            const uint64_t mc = morton_encode(w.cur.x, w.cur.y, w.cur.z);
            const uint64_t key = mc & kBlockMask;
            if (key != cached_key) { brick = find_brick(m, key); cached_key = key; }
            if (brick == kNoBrick) continue;
            const size_t at = size_t(brick) * 64 + (uint32_t(mc) & 63u);
            if (!(m.brick_cnt[at] & ~kPoisonedCount) && !m.brick_pack[at]) continue;   // counts only grow during this kernel: "holds a leaf" cannot change
            atomicAdd(reinterpret_cast<unsigned long long*>(&m.brick_sum[at]), (unsigned long long)(long long)kSumOne);   // +trunc in the map's units
            atomicAdd(&m.brick_cnt[at], 1u);
            carved++;
        } while (w.next());
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) carved += __shfl_xor(carved, o);
    if ((threadIdx.x & 63) == 0 && carved) atomicAdd(&h->n_updates[(blockIdx.x % kUpdateStripes) * 16u], carved);
}

enum { kQueryLeaf = 0, kQueryTrilinear = 1 };

__global__ void __launch_bounds__(256)
k_query_points(MapView m, const float* __restrict__ xyz, uint32_t n, uint32_t mode, float res, float recip, float trunc, float* __restrict__ sd_out,
               uint32_t* __restrict__ w_out) {
    const double metres_per_unit = double(trunc) / double(kSumOne);
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const F3 p{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        float sd = __int_as_float(0x7fc00000);
        uint32_t w = 0;
        const float lim = 1073741824.0f;   // as point_usable: voxels beyond 2^30 are nobody's
        if (finite3(p) && fabsf(p.x * recip) < lim && fabsf(p.y * recip) < lim && fabsf(p.z * recip) < lim) {
            const bool found = mode == kQueryTrilinear ? sample_trilinear(m, p, res, recip, metres_per_unit, &sd, &w) : read_leaf(m, voxel_of(p, recip), metres_per_unit, &sd, &w);
            if (!found) { sd = __int_as_float(0x7fc00000); w = 0; }
        }
        sd_out[i] = sd;
        if (w_out) w_out[i] = w;
    }
}

// t_hit[i]: distance along ray i (origin + t * direction, direction as given -- unit length expected) of the first crossing of
// the trilinear field from > 0 to <= 0 within [0, t_max]; NaN if there is none.  Samples where the field is undefined (a missing
// leaf) break the chain: a crossing needs two consecutive defined samples.
__global__ void __launch_bounds__(256)
k_raycast(MapView m, const float* __restrict__ origins, const float* __restrict__ directions, uint32_t n, float t_max, float res, float recip, float trunc,
          float* __restrict__ t_hit) {
    const double metres_per_unit = double(trunc) / double(kSumOne);
    const float dt = res * 0.5f;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const F3 o{origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]}, d{directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]};
        float hit = __int_as_float(0x7fc00000);
        if (finite3(o) && finite3(d) && t_max >= 0.0f) {
            float prev_sd = 0.0f, prev_t = 0.0f;
            bool have_prev = false;
            const uint32_t steps = uint32_t(fminf(t_max / dt, 16777216.0f)) + 1u;
            for (uint32_t k = 0; k < steps; k++) {
                const float t = float(k) * dt;
                const F3 p{o.x + d.x * t, o.y + d.y * t, o.z + d.z * t};
                float sd;
                uint32_t w;
                const float lim = 1073741824.0f;
                const bool ok = fabsf(p.x * recip) < lim && fabsf(p.y * recip) < lim && fabsf(p.z * recip) < lim && sample_trilinear(m, p, res, recip, metres_per_unit, &sd, &w) && sd == sd;
                if (ok && have_prev && prev_sd > 0.0f && sd <= 0.0f) { hit = prev_t + (t - prev_t) * (prev_sd / (prev_sd - sd)); break; }
                have_prev = ok; prev_sd = sd; prev_t = t;
            }
        }
        t_hit[i] = hit;
    }
}

}  // namespace chadhip

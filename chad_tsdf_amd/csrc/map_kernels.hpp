// map_kernels.hpp -- the sparse voxel-block map in HBM and the TSDF update kernels.
//
// Replaces Octree::insert(MortonCode) + _node_lookup (octree.hpp:31-78,187) and the weighted update of
// octree.hpp:153-164.
//
// Layout in HBM (structure of arrays, one "brick" per 4x4x4-voxel block, voxel index = low 6 code bits):
//   hash_keys[H]  u64   open-addressed table of block keys (code & ~63); EMPTY = ~0; linear probing, so a probe
//                       sequence walks consecutive 8-byte keys of one 128-byte line before touching the next
//   hash_vals[H]  u32   brick index of the key in the same slot (PENDING until the claiming lane has allocated)
//   brick_key[B]  u64   key of brick b (for export); brick_slot[B] u32 its hash slot (for O(bricks) clearing)
//   brick_sum[B*64] f32 sum of the clamped signed distances observed by each voxel
//   brick_cnt[B*64] u32 number of observations (the reference's Leaf::_weight)
// The reference keeps a running mean sd and a count w per leaf, with unit weight per observation
// (octree.hpp:161-163); (sum, cnt) is the same information with sd = sum / cnt, and it is what makes the update a
// commutative reduction: LDS partial sums, atomics across workgroups and an all-reduce across GPUs all combine
// by addition.
#pragma once
#include "device_math.hpp"
#include "frame_kernels.hpp"

namespace chadhip {

constexpr uint64_t kEmptyKey = ~0ull;
constexpr uint32_t kPending = 0xffffffffu;   // slot claimed, brick index not yet published
constexpr uint32_t kNoBrick = 0xfffffffeu;   // pool exhausted: updates to this block are dropped (overflow flag set)

struct MapView {
    uint64_t* hash_keys;
    uint32_t* hash_vals;
    uint32_t hash_mask;      // H - 1 (H a power of two)
    uint64_t* brick_key;
    uint32_t* brick_slot;    // hash slot of brick b, so that clearing the map costs O(bricks), not O(table)
    float* brick_sum;
    uint32_t* brick_cnt;
    uint32_t* brick_count;   // device counter of allocated bricks
    uint32_t brick_capacity;
};

__device__ __forceinline__ uint32_t hash_block(uint64_t key, uint32_t mask) {
    return uint32_t(((key >> 6) * 0x9E3779B97F4A7C15ull) >> 32) & mask;
}

// Brick index of `key`, creating the brick if needed.  Safe to call from divergent lanes: the claiming lane
// publishes its brick index before any lane of its wave reaches the wait loop, and lanes of other waves are
// never waited on while masked.  Returns kNoBrick when the table or pool is full.
__device__ __forceinline__ uint32_t find_or_create_brick(const MapView& m, uint64_t key, FrameHeader* h) {
    uint32_t slot = hash_block(key, m.hash_mask);
    bool claimed = false, found = false;
    for (uint32_t probes = 0; probes <= m.hash_mask; probes++) {
        const uint64_t seen = __hip_atomic_load(&m.hash_keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == key) { found = true; break; }
        if (seen == kEmptyKey) {
            const uint64_t prev = atomicCAS((unsigned long long*)&m.hash_keys[slot], (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (prev == kEmptyKey) { claimed = true; found = true; break; }
            if (prev == key) { found = true; break; }
        }
        slot = (slot + 1) & m.hash_mask;
    }
    if (!found) { atomicOr(&h->overflow, 1u); return kNoBrick; }
    if (claimed) {
        uint32_t b = atomicAdd(m.brick_count, 1u);
        if (b >= m.brick_capacity) { atomicOr(&h->overflow, 1u); b = kNoBrick; }
        else { m.brick_key[b] = key; m.brick_slot[b] = slot; }
        __hip_atomic_store(&m.hash_vals[slot], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t b;
    do { b = __hip_atomic_load(&m.hash_vals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (b == kPending);
    return b;
}

// Lookup only (no creation); kNoBrick if absent.
__device__ __forceinline__ uint32_t find_brick(const MapView& m, uint64_t key) {
    uint32_t slot = hash_block(key, m.hash_mask);
    for (uint32_t probes = 0; probes <= m.hash_mask; probes++) {
        const uint64_t seen = m.hash_keys[slot];
        if (seen == key) return m.hash_vals[slot];
        if (seen == kEmptyKey) return kNoBrick;
        slot = (slot + 1) & m.hash_mask;
    }
    return kNoBrick;
}

// ---- update, general form: one lane per point, global atomics per observation (octree.hpp:86-164) -----------
__global__ void __launch_bounds__(256)
k_update_scatter(const float4* __restrict__ pts, const float4* __restrict__ normals, FrameHeader* __restrict__ h,
                 F3 pos, float res, float recip, float trunc, MapView m, uint32_t adaptive) {
    if (adaptive && !frame_is_sparse(h)) return;  // dense frame: k_update_tiled handles it
    const uint32_t n = h->n_usable;
    unsigned long long applied = 0;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const float4 q = pts[i], nq = normals[i];
        const F3 p{q.x, q.y, q.z}, nrm{nq.x, nq.y, nq.z};
        RayWalk w;
        w.begin(p, pos, res, recip, trunc);
        uint64_t cached_key = kEmptyKey;
        uint32_t brick = kNoBrick;
        do {
            const uint64_t mc = morton_encode(w.cur.x, w.cur.y, w.cur.z);
            const uint64_t key = mc & kBlockMask;
            if (key != cached_key) { brick = find_or_create_brick(m, key, h); cached_key = key; }
            if (brick != kNoBrick) {
                const size_t at = size_t(brick) * 64 + (uint32_t(mc) & 63u);
                atomicAdd(&m.brick_sum[at], observe(w.cur, p, nrm, res, trunc));
                atomicAdd(&m.brick_cnt[at], 1u);
                applied++;
            }
        } while (w.next());
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) applied += __shfl_xor(applied, o);
    if ((threadIdx.x & 63) == 0 && applied) atomicAdd(&h->n_updates[(blockIdx.x & 31u) * 16u], applied);
}

// ---- stage entries for parity tests ----------------------------------------------------------------------------
__global__ void k_walk_rays(const float* __restrict__ xyz, uint32_t n, F3 pos, float res, float recip, float trunc,
                            int* __restrict__ voxels, uint32_t* __restrict__ counts, uint32_t stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const F3 p = load_point(xyz, i);
    uint32_t k = 0;
    if (point_usable(p, pos, recip)) {
        RayWalk w;
        w.begin(p, pos, res, recip, trunc);
        do {
            if (k < stride) { int* o = voxels + (size_t(i) * stride + k) * 3; o[0] = w.cur.x; o[1] = w.cur.y; o[2] = w.cur.z; }
            k++;
        } while (w.next());
    }
    counts[i] = k;
}

__global__ void k_morton_codes(const float* __restrict__ xyz, uint32_t n, float recip, uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const I3 v = voxel_of(load_point(xyz, i), recip);
    out[i] = morton_encode(v.x, v.y, v.z);
}

// ---- map maintenance -----------------------------------------------------------------------------------------
__global__ void k_fill_u64(uint64_t* p, size_t n, uint64_t v) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = v;
}
__global__ void k_fill_u32(uint32_t* p, size_t n, uint32_t v) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = v;
}

// Copy bricks in the order given by `order` (brick indices) into dense output arrays: one wave per brick,
// 256-byte rows, fully coalesced.
__global__ void __launch_bounds__(256)
k_gather_bricks(const uint32_t* __restrict__ order, uint32_t count, const uint64_t* __restrict__ brick_key,
                const float* __restrict__ brick_sum, const uint32_t* __restrict__ brick_cnt,
                uint64_t* __restrict__ out_key, float* __restrict__ out_sum, uint32_t* __restrict__ out_cnt) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t k = wave; k < count; k += waves) {
        const uint32_t b = order ? order[k] : k;
        out_sum[size_t(k) * 64 + lane] = brick_sum[size_t(b) * 64 + lane];
        out_cnt[size_t(k) * 64 + lane] = brick_cnt[size_t(b) * 64 + lane];
        if (lane == 0) out_key[k] = brick_key[b];
    }
}

// Insert externally produced blocks (a merged submap) into an empty map: one wave per block.
__global__ void __launch_bounds__(256)
k_import_bricks(const uint64_t* __restrict__ keys, const float* __restrict__ sum, const uint32_t* __restrict__ cnt, uint32_t count,
                FrameHeader* __restrict__ h, MapView m) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t k = wave; k < count; k += waves) {
        uint32_t b = 0;
        if (lane == 0) b = find_or_create_brick(m, keys[k], h);
        b = __shfl(b, 0);
        if (b == kNoBrick) continue;
        atomicAdd(&m.brick_sum[size_t(b) * 64 + lane], sum[size_t(k) * 64 + lane]);
        atomicAdd(&m.brick_cnt[size_t(b) * 64 + lane], cnt[size_t(k) * 64 + lane]);
    }
}

// Octree::clear() (octree.hpp:22-29) in O(bricks) and without a host round trip: the number of live bricks is read
// from device memory; one wave per brick releases its hash slot and zeroes its two 256-byte rows.
// (The counter itself is reset by k_reset_counters enqueued behind this kernel.)
__global__ void __launch_bounds__(256)
k_clear_bricks(MapView m) {
    const uint32_t blocks = min(*m.brick_count, m.brick_capacity);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t b = wave; b < blocks; b += waves) {
        if (lane == 0) {
            const uint32_t slot = m.brick_slot[b];
            m.hash_keys[slot] = kEmptyKey;
            m.hash_vals[slot] = kPending;
        }
        m.brick_sum[size_t(b) * 64 + lane] = 0.0f;
        m.brick_cnt[size_t(b) * 64 + lane] = 0u;
    }
}

__global__ void k_reset_counters(uint32_t* brick_count, FrameHeader* h, uint32_t clear_overflow) {
    *brick_count = 0u;
    if (clear_overflow) h->overflow = 0u;
}

// Rebuild the hash of a grown map: brick b (key already copied) takes a slot of the new, larger table.
__global__ void k_rehash_bricks(MapView m, uint32_t blocks, FrameHeader* __restrict__ h) {
    for (size_t b = size_t(blockIdx.x) * blockDim.x + threadIdx.x; b < blocks; b += size_t(gridDim.x) * blockDim.x) {
        const uint64_t key = m.brick_key[b];
        uint32_t slot = hash_block(key, m.hash_mask);
        for (uint32_t probes = 0; probes <= m.hash_mask; probes++) {
            if (atomicCAS((unsigned long long*)&m.hash_keys[slot], (unsigned long long)kEmptyKey, (unsigned long long)key) == kEmptyKey) {
                m.hash_vals[slot] = uint32_t(b);
                m.brick_slot[b] = slot;
                break;
            }
            slot = (slot + 1) & m.hash_mask;
        }
    }
    (void)h;
}

// keys for sorting bricks by block key: key' = brick_key (ascending), payload = brick index (implicit)
__global__ void k_copy_u64(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, uint32_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) dst[i] = src[i];
}

}  // namespace chadhip

// tile_update.hpp -- the TSDF update with the voxel neighbourhood staged in LDS (reference: octree.hpp:80-170).
//
// Points arrive grouped by 4x4x4-voxel block (the "point block").  A ray segment reaches at most
// R = ceil(trunc/res) + 1 voxels away from its point, so every voxel a point block's rays touch lies in the cube of
// D = 4 + 2R voxels around the block: D^3 x 8 bytes of LDS -- 8 KiB for trunc = 2*res (D = 10),
// 13.5 KiB for trunc = 3*res (D = 12, the largest tile kept in LDS).  One wave per work item (a point block, cut
// at every 256th point by k_frame_gather); per item:
//   0. lanes 0..26 compute the keys of the 3x3x3 neighbour blocks and LOAD their home hash slots -- speculative,
//      nothing is created -- so that the answers are back when the walk is over;
//   1. each lane walks the rays of its points and accumulates (d, 1) into the tile with ONE ds_add_u64 per
//      observation.  A tile cell is a 64-bit word [count : 24 | sum of (d * 2^30/trunc + 2^30) : 40]: the
//      fixed-point step trunc * 2^-30 is finer than the f32 rounding of d itself, an item adds at most 256
//      observations per cell (no carry into the count), and integer sums do not depend on the order of arrival.
//      The walk itself is the reference's traversal (same f32 operations on tmax/delta, same comparison tree, so
//      the voxel lists are bit-identical) with the bookkeeping folded into three registers: the LDS address and the
//      per-axis remaining step counts share one word that a step updates with a single add, and the signed
//      distance -- for which the bar is 1e-4, not bit equality -- is carried incrementally in fixed-point scale.
//      ~25 VALU instructions per voxel step (the first version: ~110).
//      Rays that leave the tile (float rounding at the rim, trunc > 3*res) or whose normal is not finite (Q4) take
//      the global path of k_update_scatter instead, outside the hot loop;
//   2. the 27 bricks of the tile are read back (lane = voxel of the brick, all reads in flight together) to learn
//      which of them received observations; lanes 0..26 whose brick was touched and whose speculative load did
//      not already show the key claim or find the slot with one CAS (map_kernels.hpp);
//   3. every touched brick is flushed with two 256-byte wave-wide atomic adds (sum, count), the full-rate shape for
//      memory-side atomics, and its part of the tile is zeroed for the next item.
// Work items are pulled from 256 separate counters (a returning atomic on ONE shared address serialises at ~25 ns
// per dequeue).  Sparse frames (few points per block) skip this kernel and take k_update_scatter instead: with one
// point per block the staging is pure overhead.
#pragma once
#include "map_kernels.hpp"

namespace chadhip {

constexpr int kTileWaves = 4;            // independent waves per workgroup, each with its own tile (no workgroup barriers)
constexpr int kTileThreads = 64 * kTileWaves;
constexpr uint32_t kMaxReach = 4;        // tiles beyond D = 12 do not fit the LDS budget per wave
constexpr float kFixedOne = 1073741824.0f;  // 2^30: |d| = trunc in the tile's fixed-point scale

// Order this wave's LDS accesses (issued in order by the hardware) without a workgroup barrier and without
// draining the global atomics still in flight, which __syncthreads() would do.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
__device__ __forceinline__ lds_u64_t* lds_cell(uint32_t byte_address) { return reinterpret_cast<lds_u64_t*>(byte_address); }

// One ray inside the tile.  `s` packs the LDS byte address of the current cell (high 16 bits) with the number of
// steps still allowed along each axis (x: bits 0-2, y: 4-6, z: 8-10, each under a guard bit 3/7/11): a step along
// axis a adds k[a] = (address stride << 16) - (1 << 4a), and the walk is over (octree.hpp:130-148: the stepped axis
// has passed the ray's last voxel) exactly when that clears the axis' guard bit.
struct LeanRay {
    float tx, ty, tz, dx, dy, dz;   // tmax / tdelta of the traversal (octree.hpp:102-118)
    float d, ddx, ddy, ddz;         // signed distance at the current voxel, times 2^30/trunc, and its change per step
    uint32_t s, kx, ky, kz;
};

// kStamps: diagnostic instantiation that accumulates s_memtime cycles per phase into `stamps` (never the product path)
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

template <bool kStamps>
__global__ void __launch_bounds__(kTileThreads, 5)  // <= 96 VGPRs: 5 workgroups per CU, what the 32 KB of LDS per workgroup allow
k_update_tiled(const float4* __restrict__ pts, const float4* __restrict__ normals, const uint4* __restrict__ items, FrameHeader* __restrict__ h,
               F3 pos, float res, float recip, float trunc, MapView m, uint32_t reach, uint32_t adaptive, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
    if (adaptive && frame_is_sparse(h)) return;
    const int R = int(reach), D = 4 + 2 * R;
    const uint32_t cells = uint32_t(D * D * D);
    const uint32_t lane = threadIdx.x & 63u, wave_in_group = threadIdx.x >> 6;
    // cells + 1 words per wave: the last one is never added to and stands in for cells outside the tile
    unsigned long long* tile = reinterpret_cast<unsigned long long*>(tile_raw) + size_t(wave_in_group) * (cells + 1u);
    const uint32_t tile_base = uint32_t(uintptr_t(tile));  // LDS byte address (low half of the flat address)
    const float to_fixed = kFixedOne / trunc, from_fixed = trunc / kFixedOne;
    for (uint32_t k = lane; k <= cells; k += 64) tile[k] = 0ull;
    const uint32_t nitems = h->n_items;
    unsigned long long applied = 0;

    // lane = voxel v of a brick (Morton order of the low 6 code bits): its cell in the centre brick, and for which of
    // the 27 bricks of the 3x3x3 neighbourhood that voxel lies inside the tile
    const int vx = int((lane & 1u) | ((lane >> 2) & 2u)), vy = int(((lane >> 1) & 1u) | ((lane >> 3) & 2u)), vz = int(((lane >> 2) & 1u) | ((lane >> 4) & 2u));
    const int centre_cell = (vx + R) + D * (vy + R) + D * D * (vz + R);
    uint32_t lane_valid = 0;
    for (int b = 0; b < 27; b++) {
        const int tx = (b % 3 - 1) * 4 + vx + R, ty = ((b / 3) % 3 - 1) * 4 + vy + R, tz = (b / 9 - 1) * 4 + vz + R;
        if (uint32_t(tx) < uint32_t(D) && uint32_t(ty) < uint32_t(D) && uint32_t(tz) < uint32_t(D)) lane_valid |= 1u << b;
    }
    // lanes 0..26: offset of neighbour brick `lane` from the point block
    const int nbx = (int(lane % 3u) - 1) * 4, nby = (int((lane / 3u) % 3u) - 1) * 4, nbz = (int(lane / 9u) - 1) * 4;
    wave_sync();

    // Work items are pulled from one of kWorkGroups counters (this wave's group = wave % kWorkGroups; group g owns
    // items g, g + kWorkGroups, ...): ~20 waves share a counter, so the load balances dynamically inside a group
    // while no counter sees more than a few dozen atomics.
    const uint32_t wave_global = blockIdx.x * kTileWaves + wave_in_group;
    const uint32_t group = wave_global % kWorkGroups;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t_begin = 0, items_done = 0;
    if (kStamps) t_begin = stamp_now();
    auto take_ticket = [&]() {  // one returning atomic per wave and item, broadcast through an SGPR
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&h->work_cursor[group * 16u], 1u);
        return uint32_t(__builtin_amdgcn_readfirstlane(int(v)));
    };
    uint32_t turn = take_ticket();
    for (;;) {
        if (kStamps) t0 = stamp_now();
        const uint32_t item = group + kWorkGroups * turn;
        if (item >= nitems) break;
        const uint4 record = items[item];  // (first point, number of points <= kItemPoints, block key), from k_frame_gather
        turn = take_ticket();              // the next item's ticket is fetched while this one is processed
        if (kStamps) items_done++;
        const uint32_t first = uint32_t(__builtin_amdgcn_readfirstlane(int(record.x))), len = uint32_t(__builtin_amdgcn_readfirstlane(int(record.y)));
        const uint64_t block_key = uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(record.z)))) | (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(record.w)))) << 32);
        const I3 corner = morton_decode(block_key);                          // lowest voxel of the point block
        const int ox = corner.x - R, oy = corner.y - R, oz = corner.z - R;    // tile origin

        // step 0: speculative look-up of the 27 neighbour blocks (plain loads that bypass the L1: a stale "empty" only
        // sends the lane to the CAS below; keys never change or vanish during an insert)
        uint64_t my_key = kEmptyKey, early = kEmptyKey;
        uint32_t my_slot = 0;
        if (lane < 27u) {
            my_key = morton_encode(corner.x + nbx, corner.y + nby, corner.z + nbz);
            my_slot = hash_block(my_key, m.hash_mask);
            early = __hip_atomic_load(&m.hash_keys[my_slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (kStamps) { const unsigned long long t1 = stamp_now(); ph[0] += t1 - t0; t0 = t1; }

        // step 1: the walk
        for (uint32_t k = lane; k < len; k += 64) {
            const float4 q = pts[first + k], nq = normals[first + k];
            const F3 p{q.x, q.y, q.z}, nrm{nq.x, nq.y, nq.z};
            RayWalk w;
            w.begin(p, pos, res, recip, trunc);
            const int lx = w.cur.x - ox, ly = w.cur.y - oy, lz = w.cur.z - oz;
            const uint32_t spx = uint32_t(abs(w.last.x - w.cur.x)), spy = uint32_t(abs(w.last.y - w.cur.y)), spz = uint32_t(abs(w.last.z - w.cur.z));
            // every voxel of the walk lies in the box spanned by its first and last voxel
            const bool inside = uint32_t(lx) < uint32_t(D) && uint32_t(ly) < uint32_t(D) && uint32_t(lz) < uint32_t(D) &&
                                uint32_t(w.last.x - ox) < uint32_t(D) && uint32_t(w.last.y - oy) < uint32_t(D) && uint32_t(w.last.z - oz) < uint32_t(D) &&
                                (spx | spy | spz) < 8u && finite3(nrm);
            if (inside) {
                LeanRay r;
                r.tx = w.tmax.x; r.ty = w.tmax.y; r.tz = w.tmax.z;
                r.dx = w.delta.x; r.dy = w.delta.y; r.dz = w.delta.z;
                const float scale = res * to_fixed;
                r.d = dot3(nrm, F3{float(w.cur.x) * res - p.x, float(w.cur.y) * res - p.y, float(w.cur.z) * res - p.z}) * to_fixed;  // octree.hpp:157-158
                r.ddx = float(w.step.x) * scale * nrm.x; r.ddy = float(w.step.y) * scale * nrm.y; r.ddz = float(w.step.z) * scale * nrm.z;
                r.s = ((tile_base + 8u * uint32_t(lx + D * ly + D * D * lz)) << 16) | 0x888u | spx | (spy << 4) | (spz << 8);
                r.kx = (uint32_t(8 * w.step.x) << 16) - 1u;
                r.ky = (uint32_t(8 * D * w.step.y) << 16) - 16u;
                r.kz = (uint32_t(8 * D * D * w.step.z) << 16) - 256u;
                do {
                    const float clamped = __builtin_amdgcn_fmed3f(r.d, -kFixedOne, kFixedOne);  // octree.hpp:159
                    const uint32_t fixed = uint32_t(int(clamped)) + (1u << 30);
                    __hip_atomic_fetch_add(lds_cell(r.s >> 16), (1ull << 40) | fixed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const bool xy = r.tx < r.ty, xz = r.tx < r.tz, yz = r.ty < r.tz;   // comparison tree of octree.hpp:126-150
                    const bool sx = xy && xz, sy = !xy && yz, sz = !(sx || sy);
                    r.tx += sx ? r.dx : 0.0f;   // tmax >= 0 (or NaN/inf): adding +0.0f leaves it unchanged
                    r.ty += sy ? r.dy : 0.0f;
                    r.tz += sz ? r.dz : 0.0f;
                    r.d += sx ? r.ddx : (sy ? r.ddy : r.ddz);
                    r.s += sx ? r.kx : (sy ? r.ky : r.kz);
                } while ((r.s & 0x888u) == 0x888u);
            } else {  // straight to HBM, one observation at a time (rare)
                uint64_t far_key = kEmptyKey;
                uint32_t far_brick = kNoBrick;
                do {
                    const uint64_t mc = morton_encode(w.cur.x, w.cur.y, w.cur.z);
                    const uint64_t key = mc & kBlockMask;
                    if (key != far_key) { far_brick = find_or_create_brick(m, key, h); far_key = key; }
                    if (far_brick != kNoBrick) {
                        const size_t at = size_t(far_brick) * 64 + (uint32_t(mc) & 63u);
                        atomicAdd(&m.brick_sum[at], observe(w.cur, p, nrm, res, trunc));
                        atomicAdd(&m.brick_cnt[at], 1u);
                        applied++;
                    }
                } while (w.next());
            }
        }
        wave_sync();  // all LDS adds of this item are issued (in order) before the read-back
        if (kStamps) { const unsigned long long t1 = stamp_now(); ph[3] += t1 - t0; t0 = t1; }

        // step 2: which of the 27 bricks received observations (all 27 reads in flight together)
        uint32_t touched = 0;
#pragma unroll 1
        for (int bz = 0; bz < 3; bz++) {  // one slab of 9 bricks at a time, its 9 reads in flight together (27 hoisted addresses cost 27 VGPRs)
            const uint32_t valid9 = lane_valid >> (9 * bz);
            const int slab = centre_cell + 4 * D * D * (bz - 1);
            uint32_t t9 = 0;
#pragma unroll
            for (int b = 0; b < 9; b++) {
                const uint32_t at = ((valid9 >> b) & 1u) ? uint32_t(slab + 4 * ((b % 3 - 1) + D * (b / 3 - 1))) : cells;
                t9 |= __ballot(tile[at] != 0ull) ? (1u << b) : 0u;
            }
            touched |= t9 << (9 * bz);
        }
        if (kStamps) { asm volatile("" :: "s"(touched)); const unsigned long long t1 = stamp_now(); ph[6] += t1 - t0; t0 = t1; }
        uint32_t my_brick = kNoBrick;
        if (lane < 27u && ((touched >> lane) & 1u)) my_brick = early == my_key ? my_slot : find_or_create_brick(m, my_key, h, my_slot);
        if (kStamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); asm volatile("" :: "v"(my_brick)); const unsigned long long t1 = stamp_now(); ph[7] += t1 - t0; t0 = t1; }

        // step 3: flush
        for (uint32_t left = touched; left;) {
            const int b = __ffs(int(left)) - 1;
            left &= left - 1u;
            const uint32_t brick = uint32_t(__builtin_amdgcn_readlane(int(my_brick), b));
            const int off = 4 * ((b % 3 - 1) + D * ((b / 3) % 3 - 1) + D * D * (b / 9 - 1));
            const uint32_t at = ((lane_valid >> b) & 1u) ? uint32_t(centre_cell + off) : cells;
            const unsigned long long cell = tile[at];
            if (cell) {
                const uint32_t c = uint32_t(cell >> 40);
                // sum of d = (sum of biased fixed values - c * 2^30) * trunc / 2^30: exact integer, one rounding to f32
                const long long fixed_sum = (long long)(cell & ((1ull << 40) - 1)) - ((long long)c << 30);
                if (brick != kNoBrick) {
                    atomicAdd(&m.brick_sum[size_t(brick) * 64 + lane], float(fixed_sum) * from_fixed);
                    atomicAdd(&m.brick_cnt[size_t(brick) * 64 + lane], c);
                    applied += c;
                }
                tile[at] = 0ull;
            }
        }
        wave_sync();
        if (kStamps) { const unsigned long long t1 = stamp_now(); ph[4] += t1 - t0; t0 = t1; }
    }
    if (kStamps && lane == 0) {
        ph[5] = stamp_now() - t_begin;
        ph[1] = t_begin;        // diagnostic build only: start time and items done replace the unused slots
        ph[2] = items_done;
        for (int k = 0; k < 8; k++) stamps[size_t(wave_global) * 8 + k] = ph[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) applied += __shfl_xor(applied, o);
    if (lane == 0 && applied) atomicAdd(&h->n_updates[(wave_global & 31u) * 16u], applied);
}

inline uint32_t tile_reach(float res, float trunc) {
    const float r = std::ceil(trunc / res) + 1.0f;
    return r < 1.0f ? 1u : (r > float(kMaxReach) ? kMaxReach : uint32_t(r));
}

inline void tile_update_enqueue(const float4* pts, const float4* normals, const uint4* items, FrameHeader* h, F3 pos, float res, float recip, float trunc,
                                const MapView& m, hipStream_t stream, bool adaptive, unsigned long long* stamps = nullptr) {
    const uint32_t reach = tile_reach(res, trunc);
    const uint32_t d = 4 + 2 * reach;
    const size_t lds = (size_t(d) * d * d + 1) * 8 * kTileWaves;
    uint32_t per_cu = uint32_t(std::min<size_t>(6, (160 * 1024) / (lds + 64)));  // persistent workgroups (of 4 waves) per CU
    if (const char* e = std::getenv("CHAD_DEBUG_WAVES_PER_CU")) per_cu = uint32_t(std::atoi(e));
    if (stamps)
        hipLaunchKernelGGL(k_update_tiled<true>, dim3(256 * per_cu), dim3(kTileThreads), lds, stream, pts, normals, items, h, pos, res, recip, trunc, m, reach, adaptive ? 1u : 0u, stamps);
    else
        hipLaunchKernelGGL(k_update_tiled<false>, dim3(256 * per_cu), dim3(kTileThreads), lds, stream, pts, normals, items, h, pos, res, recip, trunc, m, reach, adaptive ? 1u : 0u, stamps);
}

}  // namespace chadhip

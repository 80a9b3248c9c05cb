// tile_update.hpp -- the TSDF update with the voxel neighbourhood staged in LDS (reference: octree.hpp:80-170).
//
// Points arrive grouped by 4x4x4-voxel block (the "point block").  A ray segment reaches at most
// R = ceil(trunc/res) + 1 voxels away from its point, so every voxel a point block's rays touch lies in the cube of
// D = 4 + 2R voxels around the block: D^3 x 8 bytes of LDS -- 8 KiB for trunc = 2*res (D = 10),
// 13.5 KiB for trunc = 3*res (D = 12, the largest tile kept in LDS).  One wave per work item (a point block, cut
// at every 256th point by k_frame_gather):
//   1. each lane walks the rays of its points and accumulates (d, 1) into the tile with ONE ds_add_u64 per
//      observation: the LDS atomic units retire about one lane-operation per cycle and CU, so two atomics per
//      observation (f32 sum + u32 count) were half of this kernel's time.  A tile cell is a 64-bit word
//      [count : 24 | sum of (d * 2^30/trunc + 2^30) : 40]; the fixed-point step trunc * 2^-30 is finer than the
//      f32 rounding of d itself, an item adds at most 256 observations per cell (no carry into the count), and
//      integer sums do not depend on the order of arrival.  The tile is a plain x-fastest array, so a step of
//      the walk moves the LDS address by a per-ray constant.
//      A ray that leaves the tile (trunc > 3*res, or float rounding at the rim) is walked a second time
//      afterwards and its outside voxels take the global path of k_update_scatter -- kept out of the hot loop;
//   2. the tile is read back brick by brick (27 bricks of the 3x3x3 neighbourhood, lane = voxel of the brick);
//      lanes 0..26 look up / create the bricks that received observations in the HBM block hash (27 probes per
//      work item instead of one per observation);
//   3. every such brick is flushed with two 256-byte wave-wide atomic adds (sum, count), the full-rate shape for
//      memory-side atomics, and its part of the tile is zeroed for the next item.
// Work items are pulled from 256 separate counters (a returning atomic on ONE shared address serialises at ~25 ns
// per dequeue, which measured as half of this kernel's time; static round-robin left it waiting for the
// unluckiest wave).  Sparse frames (few points per block) skip this kernel and take
// k_update_scatter instead: with one point per block the staging is pure overhead.
#pragma once
#include "map_kernels.hpp"

namespace chadhip {

constexpr int kTileWaves = 4;            // independent waves per workgroup, each with its own tile (no workgroup barriers)
constexpr int kTileThreads = 64 * kTileWaves;

// Order this wave's LDS accesses (issued in order by the hardware) without a workgroup barrier and without
// draining the global atomics still in flight, which __syncthreads() would do.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
constexpr uint32_t kMaxReach = 4;        // tiles beyond D = 12 do not fit the LDS budget per wave

// Ray walk with the LDS address carried along (same traversal as RayWalk, octree.hpp:86-152).
struct TileWalk {
    RayWalk w;
    int lx, ly, lz;        // voxel position relative to the tile origin
    int at;                // lx + D*ly + D*D*lz
    int dx, dy, dz;        // address change of a step along each axis

    __device__ __forceinline__ void begin(F3 p, F3 pos, float res, float recip, float trunc, int ox, int oy, int oz, int D) {
        w.begin(p, pos, res, recip, trunc);
        lx = w.cur.x - ox; ly = w.cur.y - oy; lz = w.cur.z - oz;
        at = lx + D * ly + D * D * lz;
        dx = w.step.x; dy = w.step.y * D; dz = w.step.z * D * D;
    }
    __device__ __forceinline__ bool inside(uint32_t D) const { return uint32_t(lx) < D && uint32_t(ly) < D && uint32_t(lz) < D; }
    // Branch-free form of RayWalk::next().  Per-axis masked adds and boolean logic only: a three-way select between
    // struct members ("sx ? cur.x : sy ? cur.y : cur.z") is turned into an indexed scratch load by hipcc.
    __device__ __forceinline__ bool next() {
        const bool sx = (w.tmax.x < w.tmax.y) && (w.tmax.x < w.tmax.z);
        const bool sy = !(w.tmax.x < w.tmax.y) && (w.tmax.y < w.tmax.z);
        const bool sz = !sx && !sy;
        const int mx = sx ? w.step.x : 0, my = sy ? w.step.y : 0, mz = sz ? w.step.z : 0;
        w.cur.x += mx; w.cur.y += my; w.cur.z += mz;
        lx += mx; ly += my; lz += mz;
        at += (sx ? dx : 0) + (sy ? dy : 0) + (sz ? dz : 0);
        w.tmax.x += sx ? w.delta.x : 0.0f;   // tmax >= 0 (or NaN/inf): adding +0.0f leaves it unchanged
        w.tmax.y += sy ? w.delta.y : 0.0f;
        w.tmax.z += sz ? w.delta.z : 0.0f;
        const bool over = (sx && w.cur.x == w.last.x + w.step.x) || (sy && w.cur.y == w.last.y + w.step.y) || (sz && w.cur.z == w.last.z + w.step.z);
        return !over && (--w.budget >= 0);
    }
};

// kStamps: diagnostic instantiation that accumulates s_memtime cycles per phase into `stamps` (never the product path)
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

template <bool kStamps>
__global__ void __launch_bounds__(kTileThreads)
k_update_tiled(const float4* __restrict__ pts, const float4* __restrict__ normals, const uint64_t* __restrict__ codes,
               const uint2* __restrict__ items, FrameHeader* __restrict__ h,
               F3 pos, float res, float recip, float trunc, MapView m, uint32_t reach, uint32_t adaptive, uint32_t ablate, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
    if (adaptive && frame_is_sparse(h)) return;
    const int R = int(reach), D = 4 + 2 * R;
    const uint32_t cells = uint32_t(D * D * D);
    const uint32_t lane = threadIdx.x & 63u, wave_in_group = threadIdx.x >> 6;
    unsigned long long* tile = reinterpret_cast<unsigned long long*>(tile_raw) + size_t(wave_in_group) * cells;
    const float to_fixed = 1073741824.0f / trunc, from_fixed = trunc / 1073741824.0f;  // 2^30 / trunc
    for (uint32_t k = lane; k < cells; k += 64) tile[k] = 0ull;
    const uint32_t nitems = h->n_items;
    unsigned long long applied = 0;
    // lane = voxel v of a brick: its offset inside the brick (Morton order of the low 6 code bits)
    const int vx = int((lane & 1u) | ((lane >> 2) & 2u)), vy = int(((lane >> 1) & 1u) | ((lane >> 3) & 2u)), vz = int(((lane >> 2) & 1u) | ((lane >> 4) & 2u));
    wave_sync();

    // Work items are pulled from one of kWorkGroups counters (this wave's group = blockIdx % kWorkGroups; group g
    // owns items g, g + kWorkGroups, ...): ~20 waves share a counter, so the load balances dynamically inside a
    // group while no counter sees more than a few dozen atomics.
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
        turn = take_ticket();  // the next item's ticket is fetched while this one is processed
        if (kStamps) items_done++;
        const uint2 record = items[item];  // (first point, number of points <= kItemPoints), from k_frame_gather
        const uint32_t first = record.x, len = record.y;
        const uint64_t block = codes[first] >> 6;
        const I3 corner = morton_decode(block << 6);                          // lowest voxel of the point block
        const int ox = corner.x - R, oy = corner.y - R, oz = corner.z - R;    // tile origin
        if (kStamps) { const unsigned long long t1 = stamp_now(); ph[0] += t1 - t0; t0 = t1; }

        const uint32_t walk_len = (ablate & 1u) ? 0u : len;
        for (uint32_t k = lane; k < walk_len; k += 64) {
            const float4 q = pts[first + k], nq = normals[first + k];
            const F3 p{q.x, q.y, q.z}, nrm{nq.x, nq.y, nq.z};
            if (kStamps) { const unsigned long long t1 = stamp_now(); ph[1] += t1 - t0; t0 = t1; }
            TileWalk t;
            t.begin(p, pos, res, recip, trunc, ox, oy, oz, D);
            if (kStamps) { asm volatile("" :: "v"(t.at), "v"(t.w.tmax.x), "v"(t.w.tmax.y), "v"(t.w.tmax.z)); const unsigned long long t1 = stamp_now(); ph[2] += t1 - t0; t0 = t1; }
            bool left_tile = false;
            do {
                if (t.inside(uint32_t(D))) {
                    const float d = observe(t.w.cur, p, nrm, res, trunc);
                    if (d == d) {  // |d| <= trunc: d * 2^30/trunc fits an int32
                        const uint32_t fixed = uint32_t(__float2int_rn(d * to_fixed) + (1 << 30));
                        const int at = (ablate & 8u) ? int(lane) : t.at;  // timing experiment: conflict-free addresses
                        if (!(ablate & 16u)) atomicAdd(&tile[at], (1ull << 40) | fixed);
                        else asm volatile("" :: "v"(fixed), "v"(at));
                        applied++;
                    } else left_tile = true;  // NaN (degenerate normal, Q4): the second walk adds it in f32
                } else left_tile = true;
            } while (t.next());
            if (left_tile) {  // outside voxels straight to HBM (second walk: rare, keeps the loop above lean)
                RayWalk w;
                w.begin(p, pos, res, recip, trunc);
                uint64_t far_key = kEmptyKey;
                uint32_t far_brick = kNoBrick;
                do {
                    const uint32_t lx = uint32_t(w.cur.x - ox), ly = uint32_t(w.cur.y - oy), lz = uint32_t(w.cur.z - oz);
                    const float d = observe(w.cur, p, nrm, res, trunc);
                    if (lx < uint32_t(D) && ly < uint32_t(D) && lz < uint32_t(D) && d == d) continue;  // already in the tile
                    const uint64_t mc = morton_encode(w.cur.x, w.cur.y, w.cur.z);
                    const uint64_t key = mc & kBlockMask;
                    if (key != far_key) { far_brick = find_or_create_brick(m, key, h); far_key = key; }
                    if (far_brick != kNoBrick) {
                        const size_t at = size_t(far_brick) * 64 + (uint32_t(mc) & 63u);
                        atomicAdd(&m.brick_sum[at], d);
                        atomicAdd(&m.brick_cnt[at], 1u);
                        applied++;
                    }
                } while (w.next());
            }
        }
        wave_sync();  // all LDS adds of this item are issued (in order) before the read-back
        if (kStamps) { const unsigned long long t1 = stamp_now(); ph[3] += t1 - t0; t0 = t1; }

        // read the tile back brick by brick (lane = voxel of the brick): first only to learn which of the 27
        // bricks received observations, then again for the flush -- keeping 27 counts in registers instead cost
        // 100 VGPRs and with them most of the kernel's occupancy
        auto cell_of = [&](int b) {  // LDS index of this lane's voxel in brick b, or -1 outside the tile
            const int tx = (b % 3 - 1) * 4 + vx + R, ty = ((b / 3) % 3 - 1) * 4 + vy + R, tz = (b / 9 - 1) * 4 + vz + R;
            const bool in = uint32_t(tx) < uint32_t(D) && uint32_t(ty) < uint32_t(D) && uint32_t(tz) < uint32_t(D);
            return in ? tx + D * ty + D * D * tz : -1;
        };
        uint32_t touched = 0;
#pragma unroll 1
        for (int b = 0; b < 27; b++) {
            const int at = cell_of(b);
            const unsigned long long cell = at >= 0 ? tile[at] : 0ull;
            touched |= __ballot(cell != 0ull) ? (1u << b) : 0u;
        }
        if (kStamps) { asm volatile("" :: "s"(touched)); const unsigned long long t1 = stamp_now(); ph[6] += t1 - t0; t0 = t1; }
        // lanes 0..26: brick of neighbour (lane % 3, lane / 3 % 3, lane / 9) if it was touched
        uint32_t my_brick = kNoBrick;
        if (lane < 27u && ((touched >> lane) & 1u) && !(ablate & 4u)) {
            const int bx = int(lane % 3u) - 1, by = int((lane / 3u) % 3u) - 1, bz = int(lane / 9u) - 1;
            my_brick = find_or_create_brick(m, morton_encode(corner.x + 4 * bx, corner.y + 4 * by, corner.z + 4 * bz), h);
        }
        if (kStamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); asm volatile("" :: "v"(my_brick)); const unsigned long long t1 = stamp_now(); ph[7] += t1 - t0; t0 = t1; }
        for (uint32_t left = touched; left;) {
            const int b = __ffs(int(left)) - 1;
            left &= left - 1u;
            const uint32_t brick = uint32_t(__builtin_amdgcn_readlane(int(my_brick), b));
            const int at = cell_of(b);
            const unsigned long long cell = at >= 0 ? tile[at] : 0ull;
            if (cell) {
                const uint32_t c = uint32_t(cell >> 40);
                // sum of d = (sum of biased fixed values - c * 2^30) * trunc / 2^30: exact integer, one rounding to f32
                const long long fixed_sum = (long long)(cell & ((1ull << 40) - 1)) - ((long long)c << 30);
                if (brick != kNoBrick && !(ablate & 2u)) {
                    atomicAdd(&m.brick_sum[size_t(brick) * 64 + lane], float(fixed_sum) * from_fixed);
                    atomicAdd(&m.brick_cnt[size_t(brick) * 64 + lane], c);
                } else if (!(ablate & 6u)) applied -= c;  // dropped: pool full
                tile[at] = 0ull;
            }
        }
        wave_sync();
        if (kStamps) { const unsigned long long t1 = stamp_now(); ph[4] += t1 - t0; t0 = t1; }
    }
    if (kStamps && lane == 0) {
        ph[5] = stamp_now() - t_begin;
        ph[1] = t_begin;        // diagnostic build only: start time and items done replace the load/begin slots
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

inline void tile_update_enqueue(const float4* pts, const float4* normals, const uint64_t* codes, const uint2* items,
                                FrameHeader* h, F3 pos, float res, float recip, float trunc, const MapView& m, hipStream_t stream, bool adaptive, uint32_t ablate = 0,
                                unsigned long long* stamps = nullptr) {
    const uint32_t reach = tile_reach(res, trunc);
    const uint32_t d = 4 + 2 * reach;
    const size_t lds = size_t(d) * d * d * 8 * kTileWaves;
    uint32_t per_cu = uint32_t(std::min<size_t>(6, (160 * 1024) / (lds + 64)));  // persistent workgroups (of 4 waves) per CU
    if (const char* e = std::getenv("CHAD_DEBUG_WAVES_PER_CU")) per_cu = uint32_t(std::atoi(e));
    if (stamps)
        hipLaunchKernelGGL(k_update_tiled<true>, dim3(256 * per_cu), dim3(kTileThreads), lds, stream, pts, normals, codes, items, h, pos, res, recip, trunc, m, reach, adaptive ? 1u : 0u, ablate, stamps);
    else
        hipLaunchKernelGGL(k_update_tiled<false>, dim3(256 * per_cu), dim3(kTileThreads), lds, stream, pts, normals, codes, items, h, pos, res, recip, trunc, m, reach, adaptive ? 1u : 0u, ablate, stamps);
}

}  // namespace chadhip

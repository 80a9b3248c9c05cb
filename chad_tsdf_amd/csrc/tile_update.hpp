// tile_update.hpp -- the TSDF update with the voxel neighbourhood staged in LDS (reference: octree.hpp:80-170).
//
// Points arrive grouped by 4x4x4-voxel block (the "point block").  A ray segment reaches at most
// ceil(trunc/res) + 1 voxels away from its point, so for trunc <= 3*res every voxel a point block's rays touch
// lies in the 3x3x3 neighbourhood of bricks around it: 12^3 voxels = 27 bricks x 64 voxels x (f32 sum, u32 count)
// = 13.5 KiB of LDS.  One wave per point block:
//   1. each lane walks the rays of its points and accumulates (d, 1) into the LDS tile with ds_add_f32 / ds_add_u32
//      (an observation outside the tile -- only possible for trunc > 3*res -- takes the global path of
//      k_update_scatter instead);
//   2. lanes 0..26 look up / create the touched bricks in the HBM block hash (27 probes per point block instead
//      of one per observation);
//   3. every touched brick is flushed with two 256-byte wave-wide atomic adds (sum, count), the full-rate shape
//      for memory-side atomics, and its LDS copy is zeroed for the next point block.
// Work items are point blocks cut at every 256th point (k_frame_gather) and dealt round-robin to the persistent
// waves, so dense blocks near the sensor neither serialise in one wave nor make the tail long.
#pragma once
#include "map_kernels.hpp"

namespace chadhip {

constexpr int kTileBricks = 27;
constexpr int kTileThreads = 64;  // one wave per workgroup

__device__ __forceinline__ uint32_t voxel_in_brick(int x, int y, int z) {  // low 6 bits of the Morton code
    const uint32_t ux = uint32_t(x), uy = uint32_t(y), uz = uint32_t(z);
    return (ux & 1u) | ((uy & 1u) << 1) | ((uz & 1u) << 2) | ((ux & 2u) << 2) | ((uy & 2u) << 3) | ((uz & 2u) << 4);
}

__global__ void __launch_bounds__(kTileThreads)
k_update_tiled(const float4* __restrict__ pts, const float4* __restrict__ normals, const uint64_t* __restrict__ codes,
               const uint32_t* __restrict__ item_start, FrameHeader* __restrict__ h,
               F3 pos, float res, float recip, float trunc, MapView m, uint32_t ablate) {
    __shared__ float tile_sum[kTileBricks * 64];
    __shared__ uint32_t tile_cnt[kTileBricks * 64];
    const uint32_t lane = threadIdx.x;
    for (uint32_t k = lane; k < kTileBricks * 64; k += kTileThreads) { tile_sum[k] = 0.0f; tile_cnt[k] = 0u; }
    const uint32_t n = h->n_usable, nitems = h->n_items;
    unsigned long long applied = 0;
    __syncthreads();

    // Static round-robin over the work items: they are bounded (<= kItemPoints points) and listed in no
    // particular order, so the load evens out without a shared work counter (a returning atomic on one address
    // serialises at ~25 ns per dequeue, which was half of this kernel's time).
    {
      for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        const uint32_t first = item_start[item];
        const uint64_t block = codes[first] >> 6;
        // the item ends at the next point block, at the next multiple of kItemPoints, or at the frame's end
        const uint32_t limit = min(n, (first / kItemPoints + 1u) * kItemPoints);
        const uint32_t len = run_length(codes, first, limit, block, lane);
        const I3 corner = morton_decode(block << 6);  // lowest voxel of the point block
        const int ox = corner.x - 4, oy = corner.y - 4, oz = corner.z - 4;  // tile origin (one brick of halo)
        uint32_t touched = 0;

        for (uint32_t k = lane; k < ((ablate & 1u) ? 0u : len); k += kTileThreads) {
            const float4 q = pts[first + k], nq = normals[first + k];
            const F3 p{q.x, q.y, q.z}, nrm{nq.x, nq.y, nq.z};
            RayWalk w;
            w.begin(p, pos, res, recip, trunc);
            uint64_t far_key = kEmptyKey;
            uint32_t far_brick = kNoBrick;
            do {
                const float d = observe(w.cur, p, nrm, res, trunc);
                const uint32_t lx = uint32_t(w.cur.x - ox), ly = uint32_t(w.cur.y - oy), lz = uint32_t(w.cur.z - oz);
                if (lx < 12u && ly < 12u && lz < 12u) {
                    const uint32_t brick = (lx >> 2) + 3u * (ly >> 2) + 9u * (lz >> 2);
                    const uint32_t at = brick * 64u + voxel_in_brick(w.cur.x, w.cur.y, w.cur.z);
                    atomicAdd(&tile_sum[at], d);
                    atomicAdd(&tile_cnt[at], 1u);
                    touched |= 1u << brick;
                    applied++;
                } else {  // beyond the halo: straight to HBM
                    const uint64_t mc = morton_encode(w.cur.x, w.cur.y, w.cur.z);
                    const uint64_t key = mc & kBlockMask;
                    if (key != far_key) { far_brick = find_or_create_brick(m, key, h); far_key = key; }
                    if (far_brick != kNoBrick) {
                        const size_t at = size_t(far_brick) * 64 + (uint32_t(mc) & 63u);
                        atomicAdd(&m.brick_sum[at], d);
                        atomicAdd(&m.brick_cnt[at], 1u);
                        applied++;
                    }
                }
            } while (w.next());
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) touched |= __shfl_xor(touched, o);
        __syncthreads();  // all LDS adds of this point block are done

        // lanes 0..26: brick of neighbour (lane % 3, lane / 3 % 3, lane / 9) if it was touched
        uint32_t my_brick = kNoBrick;
        if (!(ablate & 4u) && lane < kTileBricks && ((touched >> lane) & 1u)) {
            const int bx = int(lane % 3u), by = int((lane / 3u) % 3u), bz = int(lane / 9u);
            my_brick = find_or_create_brick(m, morton_encode(ox + 4 * bx, oy + 4 * by, oz + 4 * bz), h);
        }
        uint32_t left = touched;
        while (left) {
            const uint32_t b = uint32_t(__ffs(int(left))) - 1u;
            left &= left - 1u;
            const uint32_t brick = __shfl(my_brick, int(b));
            const uint32_t at = b * 64u + lane;
            const uint32_t c = tile_cnt[at];
            if (c) {
                if (brick != kNoBrick && !(ablate & 2u)) {
                    atomicAdd(&m.brick_sum[size_t(brick) * 64 + lane], tile_sum[at]);
                    atomicAdd(&m.brick_cnt[size_t(brick) * 64 + lane], c);
                } else applied -= c;  // dropped: pool full
                tile_sum[at] = 0.0f;
                tile_cnt[at] = 0u;
            }
        }
        __syncthreads();
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) applied += __shfl_xor(applied, o);
    if (lane == 0 && applied) atomicAdd(&h->n_updates[(blockIdx.x & 31u) * 16u], applied);
}

inline void tile_update_enqueue(const float4* pts, const float4* normals, const uint64_t* codes, const uint32_t* item_start,
                                FrameHeader* h, F3 pos, float res, float recip, float trunc, const MapView& m, hipStream_t stream, uint32_t ablate = 0) {
    // persistent waves: 13.6 KiB of LDS each -> 11 per CU on 256 CUs
    hipLaunchKernelGGL(k_update_tiled, dim3(256 * 11), dim3(kTileThreads), 0, stream, pts, normals, codes, item_start, h, pos, res, recip, trunc, m, ablate);
}

}  // namespace chadhip

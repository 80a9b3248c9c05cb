// merge_kernels.hpp -- device side of the multi-GPU exchange at a submap close.
//
// The reference's leaf update is a mean of unit-weight observations (octree.hpp:161-163), so the ranks' partial (sum, count)
// bricks combine by addition; the active map holds the sums as 64-bit integers (map_kernels.hpp), so the combination is EXACT
// and independent of the number of ranks: N GPUs archive bit for bit the submap one GPU archives.  Steps (chad_hip.hip,
// merge_and_close): every rank compacts its live bricks raw (k_compact_bricks_raw) -> all-gather of the block keys -> every
// rank forms the same sorted union (radix sort + k_unique_*) and scatters its bricks into dense [U x 64] buffers
// (k_scatter_bricks_dense) -> all-reduce(sum) of the i64 sums and the u32 counts -> k_dense_to_archive converts to the closed
// submap's format exactly as the single-GPU archive does (k_compact_bricks: one rounding of the integer sum).
#pragma once
#include "map_kernels.hpp"

namespace chadhip {

// A count on the wire: the observation count in bits 0..27 and the reference's NaN-leaf flag (kPoisonedCount in the map) as
// one unit of 2^28, so that the sum over up to 15 ranks keeps both apart (a voxel observed 2^28 times in one submap is out of
// reach: that is 2.7e8 frames' worth).
constexpr uint32_t kWireCountMask = (1u << 28) - 1u;
constexpr int kMaxRanks = 15;

// Dense copy of the live bricks with the map's own 64-bit sums (k_compact_bricks converts to f32; this one does not).
__global__ void __launch_bounds__(256)
k_compact_bricks_raw(MapView m, uint32_t* __restrict__ cursor, uint32_t capacity, uint64_t* __restrict__ out_key, long long* __restrict__ out_sum,
                     uint32_t* __restrict__ out_cnt) {
    const uint32_t lane = threadIdx.x & 63;
    const bool pack_only = table_pack_only(m);
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t words = (m.hash_mask >> 5) + 1u;
    for (uint32_t w0 = wave * kWalkWords; w0 < words; w0 += waves * kWalkWords) {
        const uint32_t word = lane < kWalkWords && w0 + lane < words ? m.live_bits[w0 + lane] : 0u;
        uint64_t holders = __ballot(word != 0u);
        if (!holders) continue;
        uint32_t total = uint32_t(__popc(word));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
        uint32_t k = 0;
        if (lane == 0) k = atomicAdd(cursor, total);
        k = uint32_t(__builtin_amdgcn_readfirstlane(int(k)));
        for (; holders; holders &= holders - 1ull) {
            const uint32_t l = uint32_t(__ffsll((unsigned long long)holders)) - 1u;
            for (uint32_t bits = uint32_t(__builtin_amdgcn_readlane(int(word), int(l))); bits; bits &= bits - 1u, k++) {
                if (k >= capacity) continue;
                const uint32_t slot = (w0 + l) * 32u + uint32_t(__ffs(int(bits))) - 1u;
                long long units; uint32_t count;
                voxel_totals(m, size_t(slot) * 64 + lane, &units, &count, pack_only);
                out_sum[size_t(k) * 64 + lane] = units;
                out_cnt[size_t(k) * 64 + lane] = (count & kWireCountMask) | ((count & kPoisonedCount) ? (1u << 28) : 0u);
                if (lane == 0) out_key[k] = m.hash_keys[slot];
            }
        }
    }
}

// The ranks' key lists (gathered: rank r's keys at gathered[r * stride .. + count[r])) packed end to end.
struct RankCounts { uint32_t count[kMaxRanks + 1]; uint32_t begin[kMaxRanks + 1]; };
__global__ void __launch_bounds__(256)
k_concat_keys(const uint64_t* __restrict__ gathered, uint32_t stride, RankCounts rc, uint32_t world, uint64_t* __restrict__ out) {
    for (uint32_t r = 0; r < world; r++)
        for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < rc.count[r]; i += size_t(gridDim.x) * blockDim.x) out[rc.begin[r] + i] = gathered[size_t(r) * stride + i];
}

// ---- distinct keys of a sorted list, in order -------------------------------------------------------------------------------
constexpr uint32_t kUniqueChunk = 2048;   // keys per workgroup step (8 per thread)

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {   // 256 threads
    __shared__ uint32_t wave_sum_[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (int(lane) >= o) incl += t; }
    __syncthreads();   // (the array may still be read by the previous call)
    if (lane == 63u) wave_sum_[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; w++) before += wave_sum_[w];
    *total = wave_sum_[0] + wave_sum_[1] + wave_sum_[2] + wave_sum_[3];
    return before + incl - v;
}

// heads per chunk: chunk_heads[c] = number of i in the chunk with keys[i] != keys[i - 1] (i = 0 counts)
__global__ void __launch_bounds__(256)
k_unique_count(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ chunk_heads) {
    const uint32_t chunks = (n + kUniqueChunk - 1) / kUniqueChunk;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t j = 0; j < kUniqueChunk / 256; j++) {
            const uint32_t i = c * kUniqueChunk + j * 256 + threadIdx.x;
            if (i < n && (i == 0 || keys[i] != keys[i - 1])) mine++;
        }
        uint32_t total;
        block_exclusive_scan(mine, &total);
        if (threadIdx.x == 0) chunk_heads[c] = total;
    }
}
// exclusive scan of the chunk counts (one workgroup; a few thousand values at most), total to *total_out
__global__ void __launch_bounds__(256)
k_unique_offsets(uint32_t* __restrict__ chunk_heads, uint32_t chunks, uint32_t* __restrict__ total_out) {
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < chunks; c0 += 256) {
        const uint32_t c = c0 + threadIdx.x;
        const uint32_t v = c < chunks ? chunk_heads[c] : 0u;
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(v, &total);
        if (c < chunks) chunk_heads[c] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}
__global__ void __launch_bounds__(256)
k_unique_write(const uint64_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ chunk_offset, uint64_t* __restrict__ out) {
    const uint32_t chunks = (n + kUniqueChunk - 1) / kUniqueChunk;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        // thread t owns the 8 CONSECUTIVE keys c * 2048 + 8 t .. + 7, so that ranks follow the list's order
        const uint32_t i0 = c * kUniqueChunk + threadIdx.x * (kUniqueChunk / 256);
        uint64_t k[kUniqueChunk / 256 + 1];
        k[0] = i0 > 0 && i0 - 1 < n ? keys[i0 - 1] : ~0ull;
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t j = 0; j < kUniqueChunk / 256; j++) {
            k[j + 1] = i0 + j < n ? keys[i0 + j] : ~0ull;
            if (i0 + j < n && (i0 + j == 0 || k[j + 1] != k[j])) mine++;
        }
        uint32_t total;
        uint32_t at = chunk_offset[c] + block_exclusive_scan(mine, &total);
#pragma unroll
        for (uint32_t j = 0; j < kUniqueChunk / 256; j++)
            if (i0 + j < n && (i0 + j == 0 || k[j + 1] != k[j])) out[at++] = k[j + 1];
    }
}

// This rank's bricks into the dense buffers of the union (zeroed beforehand): one wave per brick, a binary search of its key.
__global__ void __launch_bounds__(256)
k_scatter_bricks_dense(const uint64_t* __restrict__ keys, const long long* __restrict__ sum, const uint32_t* __restrict__ cnt, uint32_t blocks,
                       const uint64_t* __restrict__ union_keys, uint32_t union_blocks, long long* __restrict__ dense_sum, uint32_t* __restrict__ dense_cnt) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t b = wave; b < blocks; b += waves) {
        const uint64_t key = keys[b];
        uint32_t lo = 0, hi = union_blocks;   // first index with union_keys[index] >= key
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (union_keys[mid] < key) lo = mid + 1; else hi = mid; }
        if (lo >= union_blocks || union_keys[lo] != key) continue;   // cannot happen: the union holds every rank's keys
        dense_sum[size_t(lo) * 64 + lane] = sum[size_t(b) * 64 + lane];
        dense_cnt[size_t(lo) * 64 + lane] = cnt[size_t(b) * 64 + lane];
    }
}

// The reduced dense buffers as a closed submap: exactly k_compact_bricks' conversion (one rounding of the exact integer sum;
// NaN where any rank flagged the reference's NaN leaf).
__global__ void __launch_bounds__(256)
k_dense_to_archive(const long long* __restrict__ dense_sum, const uint32_t* __restrict__ dense_cnt, size_t cells, float trunc, float* __restrict__ out_sum,
                   uint32_t* __restrict__ out_cnt) {
    const double metres_per_unit = double(trunc) / double(kSumOne);
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < cells; i += size_t(gridDim.x) * blockDim.x) {
        const uint32_t wire = dense_cnt[i];
        out_sum[i] = (wire >> 28) ? __int_as_float(0x7fc00000) : float(double(dense_sum[i]) * metres_per_unit);
        out_cnt[i] = wire & kWireCountMask;
    }
}

}  // namespace chadhip

// radix_sort.hpp -- stable LSD radix sort of (u64 key, u32 payload) pairs for gfx950, ascending.
//
// Replaces std::sort in sort_morton_vector (morton.hpp:81-102).  Stable, so equal keys keep their input
// order (the deterministic tie order D3 of the oracle).  8-bit digits; one pass =
//   tile_histogram (LDS digit histogram per 256-thread tile)
//   -> scan_histograms (exclusive scan of the digit-major [256][tiles] table)
//   -> scatter (wave64 ballot ranking inside each wave, per-wave digit counters in LDS across waves).
// Only the digits that can differ are sorted: the caller passes the number of significant key bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chadhip {

constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kSortThreads = 256;                       // 4 waves
constexpr int kSortItems = 8;                           // keys per thread
constexpr int kSortTile = kSortThreads * kSortItems;    // 2048 keys per workgroup
constexpr int kSortWaves = kSortThreads / 64;

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Element (wave w, step j, lane l) of tile t sits at t*kSortTile + w*(64*kSortItems) + j*64 + l: consecutive
// lanes read consecutive keys, and the tile order (w, j, l) is the input order, which keeps the sort stable.
__device__ __forceinline__ size_t tile_index(uint32_t tile, uint32_t wave, uint32_t j, uint32_t lane) {
    return size_t(tile) * kSortTile + wave * (64 * kSortItems) + j * 64 + lane;
}

// If *pass_limit (device memory) <= pass, the pass is a no-op: lets the host enqueue a fixed number of passes
// without knowing how many key bits the frame needs.  pass_limit may be null.
__global__ void __launch_bounds__(kSortThreads)
k_radix_tile_histogram(const uint64_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t ntiles,
                       uint32_t* __restrict__ tile_hist, const uint32_t* __restrict__ pass_limit, uint32_t pass) {
    if (pass_limit && pass >= *pass_limit) return;
    __shared__ uint32_t hist[kRadix];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (uint32_t j = 0; j < kSortItems; j++) {
        const size_t i = tile_index(blockIdx.x, wave, j, lane);
        if (i < n) atomicAdd(&hist[uint32_t(keys[i] >> shift) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    tile_hist[size_t(threadIdx.x) * ntiles + blockIdx.x] = hist[threadIdx.x];  // digit-major
}

// Exclusive scan of `count` counters in place by ONE workgroup of 1024 threads (count = 256 * tiles).
__global__ void __launch_bounds__(1024)
k_radix_scan(uint32_t* __restrict__ data, uint32_t count, const uint32_t* __restrict__ pass_limit, uint32_t pass) {
    if (pass_limit && pass >= *pass_limit) return;
    __shared__ uint32_t partial[1024];
    const uint32_t per = (count + 1023) / 1024;
    const uint32_t beg = threadIdx.x * per;
    const uint32_t end = beg + per < count ? beg + per : count;
    uint32_t sum = 0;
    for (uint32_t i = beg; i < end; i++) sum += data[i];
    partial[threadIdx.x] = sum;
    __syncthreads();
    // Hillis-Steele over the 1024 partials
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? partial[threadIdx.x - off] : 0;
        __syncthreads();
        partial[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = partial[threadIdx.x] - sum;  // exclusive prefix of this thread's chunk
    for (uint32_t i = beg; i < end; i++) {
        const uint32_t v = data[i];
        data[i] = run;
        run += v;
    }
}

// idx_in may be null: the payload is then the element's own position (first pass).
__global__ void __launch_bounds__(kSortThreads)
k_radix_scatter(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ idx_in,
                uint64_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out,
                uint32_t n, uint32_t shift, uint32_t ntiles, const uint32_t* __restrict__ tile_offset,
                const uint32_t* __restrict__ pass_limit, uint32_t pass) {
    if (pass_limit && pass >= *pass_limit) return;
    __shared__ uint32_t wave_count[kSortWaves][kRadix];  // then: exclusive base of (wave, digit) in the output
    for (uint32_t k = threadIdx.x; k < kSortWaves * kRadix; k += kSortThreads) (&wave_count[0][0])[k] = 0;
    __syncthreads();

    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    uint64_t key[kSortItems];
    uint32_t val[kSortItems], rank[kSortItems];
#pragma unroll
    for (uint32_t j = 0; j < kSortItems; j++) {
        const size_t i = tile_index(blockIdx.x, wave, j, lane);
        const bool live = i < n;
        key[j] = live ? keys_in[i] : ~0ull;
        val[j] = live ? (idx_in ? idx_in[i] : uint32_t(i)) : 0u;
        const uint32_t digit = uint32_t(key[j] >> shift) & (kRadix - 1);
        // lanes of this wave holding the same digit (ballot per digit bit)
        uint64_t same = __ballot(live);
#pragma unroll
        for (uint32_t b = 0; b < kRadixBits; b++) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t vote = __ballot(bit);
            same &= bit ? vote : ~vote;
        }
        uint32_t before = 0;
        if (live) {
            const uint32_t leader = uint32_t(__ffsll((unsigned long long)same)) - 1u;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&wave_count[wave][digit], uint32_t(__popcll(same)));
            base = __shfl(base, int(leader));
            before = base + uint32_t(__popcll(same & lanes_below));
        }
        rank[j] = before;  // rank among this wave's keys of the same digit, in (j, lane) order
    }
    __syncthreads();
    {   // digit d: turn the per-wave counts into output bases (tile offset + counts of earlier waves)
        const uint32_t d = threadIdx.x;
        uint32_t run = tile_offset[size_t(d) * ntiles + blockIdx.x];
#pragma unroll
        for (uint32_t w = 0; w < kSortWaves; w++) {
            const uint32_t c = wave_count[w][d];
            wave_count[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kSortItems; j++) {
        const size_t i = tile_index(blockIdx.x, wave, j, lane);
        if (i < n) {
            const uint32_t digit = uint32_t(key[j] >> shift) & (kRadix - 1);
            const uint32_t pos = wave_count[wave][digit] + rank[j];
            keys_out[pos] = key[j];
            idx_out[pos] = val[j];
        }
    }
}

struct RadixSortBuffers {
    uint64_t* keys[2];
    uint32_t* idx[2];
    uint32_t* tile_hist;  // 256 * ceil(capacity / kSortTile) counters
};

inline uint32_t radix_tiles(size_t n) { return uint32_t((n + kSortTile - 1) / kSortTile); }

// Enqueue `passes` passes over key bits [0, 8*passes).  Input in keys[0] (idx[0] unused on the first pass);
// the result of pass p lands in buffer (p+1)&1.  pass_limit: optional device word; passes >= *pass_limit do nothing,
// so the sorted data is in buffer (*pass_limit & 1).
inline void radix_sort_enqueue(const RadixSortBuffers& b, uint32_t n, uint32_t passes, const uint32_t* pass_limit, hipStream_t stream) {
    if (n == 0) return;
    const uint32_t ntiles = radix_tiles(n);
    for (uint32_t p = 0; p < passes; p++) {
        const int src = p & 1, dst = src ^ 1;
        hipLaunchKernelGGL(k_radix_tile_histogram, dim3(ntiles), dim3(kSortThreads), 0, stream, b.keys[src], n, p * kRadixBits, ntiles, b.tile_hist, pass_limit, p);
        hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, stream, b.tile_hist, uint32_t(kRadix) * ntiles, pass_limit, p);
        hipLaunchKernelGGL(k_radix_scatter, dim3(ntiles), dim3(kSortThreads), 0, stream, b.keys[src], p == 0 ? nullptr : b.idx[src], b.keys[dst], b.idx[dst], n,
                           p * kRadixBits, ntiles, b.tile_hist, pass_limit, p);
    }
}

}  // namespace chadhip

// exchange_kernels.hpp -- device side of the SPARSE multi-GPU exchange at a submap close ("reduce-scatter by key
// range so each rank finalises 1/G of the blocks").  Used by frame-sharded maps that keep closed submaps as their DAG only
// (CHAD_FLAG_DAG_ONLY: BASELINE config 5); the dense all-reduce of merge_kernels.hpp stays for sharded maps that keep the archives.
//
// What travels is what exists: per block its key and a 64-bit mask of the voxels that hold observations, per such voxel ONE word when
// the rank's table only ever took the scatter form (the staging word [count : 24 | sum : 40] as it stands in the table: sparse frames)
// or two (the exact i64 sum, the u32 count) otherwise -- about 11 bytes per voxel of a cfg5 submap instead of the 768 bytes per block of
// the union that the dense exchange reduces.  Ownership is by Morton key RANGE: G - 1 splitters, drawn from samples of every rank's
// sorted keys, cut the key space so that every rank owns about 1/G of the blocks; a rank's blocks are sorted by key, so what it
// sends to rank d is one contiguous slice.  The owner adds up what it receives -- integer sums: exactly what one GPU's table would
// hold -- and turns ITS range into cluster words (submap.hpp:70-99, the conversion of k_sorted_clusters bit for bit); the ranges,
// concatenated in rank order on rank 0, are the submap's cluster words in Morton order, and the fold (Submap::finalize's other half,
// dag_device.hpp) runs there ONCE.
//
// Slice (source -> destination), nb blocks, w payload words:   keys u64[nb] | masks u64[nb] | first u32[nb] (padded to 8 bytes) |
// payload u64[w]: the words of block i are payload[first[i] * width ...], one group per set bit of masks[i], in voxel order.
#pragma once
#include "exchange_plan.hpp"
#include "merge_kernels.hpp"

namespace chadhip {

__host__ __device__ inline size_t xchg_pad8(size_t bytes) { return (bytes + 7) & ~size_t(7); }

// the live slots in key order: slot_sorted[r] = slot of the r-th smallest key (order: the sorter's payload, positions in k_table_keys' list)
__global__ void __launch_bounds__(256)
k_xchg_order(const uint32_t* __restrict__ order, const uint32_t* __restrict__ slots, uint32_t nb, uint32_t* __restrict__ slot_sorted) {
    for (size_t r = size_t(blockIdx.x) * blockDim.x + threadIdx.x; r < nb; r += size_t(gridDim.x) * blockDim.x) slot_sorted[r] = slots[order[r]];
}
// what a rank tells the others first: head[0] = its block count (low 32 bits) and payload width (bit 32: two words per voxel), then
// kXSamples keys at even distances through its sorted list (~0 where it has none)
__global__ void k_xchg_head(const uint64_t* __restrict__ keys_sorted, uint32_t nb, uint32_t wide, uint64_t* __restrict__ head) {
    const uint32_t i = threadIdx.x;
    if (i == 0) head[0] = uint64_t(nb) | (uint64_t(wide) << 32);
    if (i < kXSamples) head[1 + i] = nb ? keys_sorted[size_t(i + 1) * nb / (kXSamples + 1)] : ~0ull;
}
// range d = keys in [split[d - 1], split[d]): first_block[d] = first index whose key is >= split[d - 1]; first_block[0] = 0, first_block[world] = nb
__global__ void k_xchg_ranges(const uint64_t* __restrict__ keys_sorted, uint32_t nb, const uint64_t* __restrict__ split, uint32_t world, uint32_t* __restrict__ first_block) {
    const uint32_t d = threadIdx.x;
    if (d > world) return;
    if (d == 0) { first_block[0] = 0; return; }
    if (d == world) { first_block[world] = nb; return; }
    const uint64_t bound = split[d - 1];
    uint32_t lo = 0, hi = nb;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys_sorted[mid] < bound) lo = mid + 1; else hi = mid; }
    first_block[d] = lo;
}
// which voxels of the r-th block hold observations (a wave per brick, four bricks' rows in flight)
__global__ void __launch_bounds__(256)
k_xchg_masks(MapView m, const uint32_t* __restrict__ slot_sorted, uint32_t nb, uint64_t* __restrict__ mask) {
    const bool pack_only = table_pack_only(m);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t r0 = wave * 4u; r0 < nb; r0 += waves * 4u) {
        long long units[4]; uint32_t cnt[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) { cnt[i] = 0; if (r0 + i < nb) voxel_totals(m, size_t(slot_sorted[r0 + i]) * 64 + lane, &units[i], &cnt[i], pack_only); }
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint64_t bits = __ballot(cnt[i] != 0u);
            if (r0 + i < nb && lane == 0) mask[r0 + i] = bits;
        }
    }
}
// exclusive prefix of the masks' bit counts: chunk totals | (k_unique_offsets: scan of the totals) | prefix[r], prefix[nb] = all
__global__ void __launch_bounds__(256)
k_xchg_chunk_sums(const uint64_t* __restrict__ mask, uint32_t nb, uint32_t* __restrict__ chunk_sums) {
    const uint32_t chunks = (nb + kUniqueChunk - 1) / kUniqueChunk;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t j = 0; j < kUniqueChunk / 256; j++) { const uint32_t r = c * kUniqueChunk + j * 256 + threadIdx.x; if (r < nb) mine += uint32_t(__popcll(mask[r])); }
        uint32_t total;
        block_exclusive_scan(mine, &total);
        if (threadIdx.x == 0) chunk_sums[c] = total;
    }
}
__global__ void __launch_bounds__(256)
k_xchg_prefix(const uint64_t* __restrict__ mask, uint32_t nb, const uint32_t* __restrict__ chunk_offset, const uint32_t* __restrict__ total, uint32_t* __restrict__ prefix) {
    const uint32_t chunks = (nb + kUniqueChunk - 1) / kUniqueChunk;
    if (blockIdx.x == 0 && threadIdx.x == 0) prefix[nb] = nb ? *total : 0u;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t r0 = c * kUniqueChunk + threadIdx.x * (kUniqueChunk / 256);   // eight consecutive blocks per thread
        uint32_t bits[kUniqueChunk / 256], mine = 0;
#pragma unroll
        for (uint32_t j = 0; j < kUniqueChunk / 256; j++) { bits[j] = r0 + j < nb ? uint32_t(__popcll(mask[r0 + j])) : 0u; mine += bits[j]; }
        uint32_t all;
        uint32_t at = chunk_offset[c] + block_exclusive_scan(mine, &all);
#pragma unroll
        for (uint32_t j = 0; j < kUniqueChunk / 256; j++) { if (r0 + j < nb) prefix[r0 + j] = at; at += bits[j]; }
    }
}
// what goes to each rank: row[2 d] = blocks, row[2 d + 1] = payload words (voxels) of range d
__global__ void k_xchg_row(const uint32_t* __restrict__ first_block, const uint32_t* __restrict__ prefix, uint32_t world, uint64_t* __restrict__ row) {
    const uint32_t d = threadIdx.x;
    if (d >= world) return;
    row[2 * d] = first_block[d + 1] - first_block[d];
    row[2 * d + 1] = prefix[first_block[d + 1]] - prefix[first_block[d]];
}

struct XchgSlices { unsigned long long off[kMaxRanks + 1]; };   // byte offset of the slice for each destination in the send buffer

// the slices themselves: the bricks are read a second time (their masks fixed the layout)
__global__ void __launch_bounds__(256)
k_xchg_pack(MapView m, const uint32_t* __restrict__ slot_sorted, const uint64_t* __restrict__ keys_sorted, const uint64_t* __restrict__ mask, const uint32_t* __restrict__ prefix,
            const uint32_t* __restrict__ first_block, uint32_t nb, uint32_t world, uint32_t wide, char* __restrict__ send, XchgSlices slices) {
    const bool pack_only = table_pack_only(m);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), waves = gridDim.x * (blockDim.x >> 6);
    // sixty-four blocks per turn: every lane gathers ITS block's place in the slices (destination, index there, first payload word) and
    // writes its header words; then the wave copies the blocks' voxels one block after the other, a lane per voxel (a wave doing all of
    // that for one block at a time spent its time in half a dozen dependent loads per block)
    for (uint32_t r0 = wave * 64u; r0 < nb; r0 += waves * 64u) {
        const uint32_t r = r0 + lane;
        uint32_t slot = 0, w_lo = 0, w_hi = 0, mask_lo = 0, mask_hi = 0;
        if (r < nb) {
            uint32_t d = 0;
            for (uint32_t k = 1; k < world; k++) d = r >= first_block[k] ? k : d;
            const uint32_t first = first_block[d], n_d = first_block[d + 1] - first, i = r - first;
            char* base = send + slices.off[d];
            uint64_t* keys = reinterpret_cast<uint64_t*>(base);
            uint64_t* masks = keys + n_d;
            uint32_t* firsts = reinterpret_cast<uint32_t*>(masks + n_d);
            const uint64_t bits = mask[r];
            const uint32_t w0 = prefix[r] - prefix[first];
            keys[i] = keys_sorted[r]; masks[i] = bits; firsts[i] = w0;
            const uint64_t word0 = (uint64_t(base - send) + 16 * size_t(n_d) + xchg_pad8(4 * size_t(n_d))) / 8 + uint64_t(w0) * (wide ? 2u : 1u);   // the block's first payload word, counted from `send`
            slot = slot_sorted[r]; w_lo = uint32_t(word0); w_hi = uint32_t(word0 >> 32); mask_lo = uint32_t(bits); mask_hi = uint32_t(bits >> 32);
        }
        uint64_t* words = reinterpret_cast<uint64_t*>(send);
        const uint32_t here = min(64u, nb - r0);
        for (uint32_t j = 0; j < here; j++) {
            const uint64_t bits = uint64_t(uint32_t(__builtin_amdgcn_readlane(int(mask_lo), int(j)))) | (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(mask_hi), int(j)))) << 32);
            if (!((bits >> lane) & 1ull)) continue;
            const uint64_t word0 = uint64_t(uint32_t(__builtin_amdgcn_readlane(int(w_lo), int(j)))) | (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(w_hi), int(j)))) << 32);
            const size_t at = size_t(uint32_t(__builtin_amdgcn_readlane(int(slot), int(j)))) * 64 + lane;
            const size_t k = uint32_t(__popcll(bits & ((1ull << lane) - 1ull)));
            if (wide) {
                long long units; uint32_t cnt;
                voxel_totals(m, at, &units, &cnt, pack_only);
                words[word0 + 2 * k] = uint64_t(units); words[word0 + 2 * k + 1] = cnt;
            } else words[word0 + k] = m.brick_pack[at];   // (a pack-only table: the staging word is all there is)
        }
    }
}

// A received slice added into the owner's dense accumulators over the union of its range's keys.  One launch per source, one after the
// other on the stream: plain read-modify-write (a key occurs once per slice).  Integer sums: exactly what one table would hold.
__global__ void __launch_bounds__(256)
k_xchg_accumulate(const char* __restrict__ slice, uint32_t n_s, uint32_t wide, const uint64_t* __restrict__ union_keys, uint32_t n_union,
                  long long* __restrict__ acc_sum, uint32_t* __restrict__ acc_cnt) {
    const uint64_t* keys = reinterpret_cast<const uint64_t*>(slice);
    const uint64_t* masks = keys + n_s;
    const uint32_t* firsts = reinterpret_cast<const uint32_t*>(masks + n_s);
    const uint64_t* payload = reinterpret_cast<const uint64_t*>(slice + 16 * size_t(n_s) + xchg_pad8(4 * size_t(n_s)));
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), waves = gridDim.x * (blockDim.x >> 6);
    // sixty-four blocks per turn: every lane looks ITS block up in the union (a wave searching for one block at a time spent ten
    // microseconds per block in twenty dependent loads -- 0.74 ms per million-block slice, rocprofv3), then the wave adds the blocks' voxels
    // one block after the other, a lane per voxel
    for (uint32_t b0 = wave * 64u; b0 < n_s; b0 += waves * 64u) {
        const uint32_t mine = b0 + lane;
        uint32_t row = ~0u, first = 0, mask_lo = 0, mask_hi = 0;
        if (mine < n_s) {
            const uint64_t key = keys[mine], bits = masks[mine];
            uint32_t lo = 0, hi = n_union;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (union_keys[mid] < key) lo = mid + 1; else hi = mid; }
            if (lo < n_union && union_keys[lo] == key) row = lo;   // (always: the union holds every slice's keys)
            first = firsts[mine]; mask_lo = uint32_t(bits); mask_hi = uint32_t(bits >> 32);
        }
        const uint32_t here = min(64u, n_s - b0);
        for (uint32_t j = 0; j < here; j++) {
            const uint32_t r = uint32_t(__builtin_amdgcn_readlane(int(row), int(j)));
            const uint64_t bits = uint64_t(uint32_t(__builtin_amdgcn_readlane(int(mask_lo), int(j)))) | (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(mask_hi), int(j)))) << 32);
            const uint32_t f = uint32_t(__builtin_amdgcn_readlane(int(first), int(j)));
            if (r == ~0u || !((bits >> lane) & 1ull)) continue;
            const size_t at_word = size_t(f) + uint32_t(__popcll(bits & ((1ull << lane) - 1ull)));
            long long units; uint32_t cnt;
            if (wide) { units = (long long)payload[2 * at_word]; cnt = uint32_t(payload[2 * at_word + 1]); }
            else unpack_staged(payload[at_word], &units, &cnt);
            const size_t at = size_t(r) * 64 + lane;
            acc_sum[at] += units;
            const uint32_t before = acc_cnt[at];
            acc_cnt[at] = (((before & ~kPoisonedCount) + (cnt & ~kPoisonedCount)) & ~kPoisonedCount) | ((before | cnt) & kPoisonedCount);
        }
    }
}

// the accumulated range as cluster words: k_sorted_clusters' conversion, bit for bit (one rounding of the exact integer sum, then cluster.hpp:13-31)
__global__ void __launch_bounds__(256)
k_xchg_clusters(const long long* __restrict__ acc_sum, const uint32_t* __restrict__ acc_cnt, uint32_t n_union, float trunc, float trunc_recip,
                uint64_t* __restrict__ out_tsdf, uint64_t* __restrict__ out_weight, uint8_t* __restrict__ out_mask) {
    const double metres_per_unit = double(trunc) / double(kSumOne);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t r = wave; r < n_union; r += waves) {
        const long long units = acc_sum[size_t(r) * 64 + lane];
        const uint32_t cnt = acc_cnt[size_t(r) * 64 + lane];
        const float sum = (cnt & kPoisonedCount) ? __int_as_float(0x7fc00000) : float(double(units) * metres_per_unit);
        uint32_t sd_bits, w_bits;
        cluster_bytes(sum, cnt & ~kPoisonedCount, trunc_recip, &sd_bits, &w_bits);
        cluster_store(lane, size_t(r), sd_bits, w_bits, (cnt & ~kPoisonedCount) != 0u, out_tsdf, out_weight, out_mask);
    }
}


// ---- a range's cluster words on their way to rank 0: only the octets that hold leaves travel (a cfg5 block holds leaves in about three
// of its eight octets).  Range slice, U blocks, W words:   keys u64[U] | first u32[U] (padded to 8 bytes) | masks u8[U] (padded) |
// tsdf words u64[W] | weight words u64[W] (CHAD_FLAG_DAG_SATURATE_WEIGHTS only); block b's words are [first[b], first[b] + popcount(masks[b])).
__global__ void __launch_bounds__(256)
k_xchg_widen(const uint8_t* __restrict__ mask8, uint32_t n, uint64_t* __restrict__ mask64) {
    for (size_t b = size_t(blockIdx.x) * blockDim.x + threadIdx.x; b < n; b += size_t(gridDim.x) * blockDim.x) mask64[b] = mask8[b];
}
// rows of eight words -> the words of the set octets, in block and octet order; also the slice's `first` and mask arrays, and [U, W] for the other ranks
__global__ void __launch_bounds__(256)
k_xchg_compact_rows(const uint64_t* __restrict__ rows_tsdf, const uint64_t* __restrict__ rows_weight, const uint8_t* __restrict__ mask8, const uint32_t* __restrict__ first /* [U + 1] */,
                    uint32_t n, char* __restrict__ out, uint64_t* __restrict__ sizes /* [2] */) {
    uint32_t* out_first = reinterpret_cast<uint32_t*>(out + 8 * size_t(n));
    uint8_t* out_mask = reinterpret_cast<uint8_t*>(out + 8 * size_t(n) + xchg_pad8(4 * size_t(n)));
    uint64_t* out_words = reinterpret_cast<uint64_t*>(out + 8 * size_t(n) + xchg_pad8(4 * size_t(n)) + xchg_pad8(n));
    const uint32_t total = first[n];
    if (blockIdx.x == 0 && threadIdx.x == 0) { sizes[0] = n; sizes[1] = total; }
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < size_t(n) * 8; i += size_t(gridDim.x) * blockDim.x) {
        const uint32_t b = uint32_t(i >> 3), o = uint32_t(i & 7u), bits = mask8[b];
        if (o == 0) { out_first[b] = first[b]; out_mask[b] = uint8_t(bits); }
        if (!((bits >> o) & 1u)) continue;
        const size_t at = size_t(first[b]) + uint32_t(__popc(bits & ((1u << o) - 1u)));
        out_words[at] = rows_tsdf[i];
        if (rows_weight) out_words[size_t(total) + at] = rows_weight[i];
    }
}
// ... and back into rows on rank 0 (an octet without leaves is all ones: cluster.hpp:28-31)
__global__ void __launch_bounds__(256)
k_xchg_expand_rows(const char* __restrict__ slice, uint32_t n, uint32_t words, uint32_t weights, uint64_t* __restrict__ rows_tsdf, uint64_t* __restrict__ rows_weight, uint8_t* __restrict__ mask8) {
    const uint32_t* first = reinterpret_cast<const uint32_t*>(slice + 8 * size_t(n));
    const uint8_t* in_mask = reinterpret_cast<const uint8_t*>(slice + 8 * size_t(n) + xchg_pad8(4 * size_t(n)));
    const uint64_t* in_words = reinterpret_cast<const uint64_t*>(slice + 8 * size_t(n) + xchg_pad8(4 * size_t(n)) + xchg_pad8(n));
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < size_t(n) * 8; i += size_t(gridDim.x) * blockDim.x) {
        const uint32_t b = uint32_t(i >> 3), o = uint32_t(i & 7u), bits = in_mask[b];
        if (o == 0) mask8[b] = uint8_t(bits);
        const bool there = (bits >> o) & 1u;
        const size_t at = size_t(first[b]) + uint32_t(__popc(bits & ((1u << o) - 1u)));
        rows_tsdf[i] = there && at < words ? in_words[at] : ~0ull;
        if (weights) rows_weight[i] = there && at < words ? in_words[size_t(words) + at] : ~0ull;
    }
}

}  // namespace chadhip

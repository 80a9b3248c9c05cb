// dag_device.hpp -- Submap::finalize (submap.hpp:10-106) with the map-wide DAG of NodeLevels (levels.hpp:141-199) RESIDENT ON THE DEVICE.
//
// The reference adds every node of a submap's octree, child first, to a per-level store that keeps each distinct value once and
// hands out addresses in order of first occurrence (NodeLevel::add levels.hpp:57-88, LeafClusterLevel::add levels.hpp:123-139).
// The levels are independent stores: the addresses a level hands out depend only on the SEQUENCE of values added to it, and a
// node's value only on the addresses of its children one level down.  So the fold runs bottom-up, one level at a time, and every
// level is a batch problem: given the level's values in the reference's order of addition (ascending Morton order of the nodes,
// the signed-distance tree's node before the weight tree's), find for each value the address of its first occurrence -- in the
// store if the value is already there from an earlier submap, else new, with new addresses handed out in sequence order.
//   1. probe: every candidate looks its value up in the level's open-addressed table (u32 slots: 0 free, else the address of a
//      stored value, compared by CONTENT, never by hash alone).  A candidate that reaches a free slot claims it with
//      (the store's size before the batch) + its index -- a number no stored value's address can be, so addresses keep all 32 bits,
//      as in the reference -- and a candidate that meets a pending slot of equal content lowers the index (atomicMin): after the
//      pass every slot claimed in this batch names the batch's FIRST candidate of that content.
//   2. the candidates that own their slot are the level's new values: an exclusive scan of their word counts in sequence order
//      gives their addresses behind the store's end -- the order of first occurrence, as the reference hands them out;
//   3. commit: they append their words to the store and put their address into their slot;  4. resolve: everybody reads its slot.
// The host half this replaces (dag.hpp: dag_add_submap, a sequential fold of the cluster stream) needs 0.8 ms for a cfg2 frame
// and 0.6 s for a submap of BASELINE config 5 (1.65 M blocks, 4.8 M cluster words) on one EPYC core: at one roll-over per eleven
// frames that is 55 ms per frame, against 0.2 ms for the frame itself.  dag.hpp stays: it is the host's READ side (dag_walk,
// DagCursor over the mirrored arrays) and the fold of CHAD_DEBUG_DAG_HOST=1, which the tests hold this file against word for word.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "alloc.hpp"
#include "dag.hpp"
#include "merge_kernels.hpp"

namespace chadhip {

// (a table slot >= the store's size at the start of the batch is pending: claimed during the batch by candidate slot - size)
constexpr uint32_t kDagChunk = 2048;            // candidates per workgroup step: 256 threads x 8 CONSECUTIVE candidates (sequence order matters)
constexpr uint32_t kDagPer = kDagChunk / 256;
constexpr uint32_t kDagNone = 0xffffffffu;

struct DagScalars {   // one per level, device-resident (mirrored to the host after every fold)
    uint32_t size;        // words of the level's store in use (address 0 is reserved: levels.hpp:52, :118)
    uint32_t batch_base;  // size before the current batch's new values
    uint32_t entries;     // values stored (= table slots in use)
    uint32_t pad;
    unsigned long long uniques, dupes;   // levels.hpp:90, :141 (_uniques_n / _dupes_n)
};

struct DagLevelDev {
    uint32_t* table = nullptr;
    uint32_t table_slots = 0;      // power of two
    void* raw = nullptr;           // u32 words (node levels) or u64 words (cluster level)
    uint32_t raw_capacity = 0;     // words
};

__device__ __forceinline__ uint64_t dag_mix_dev(uint64_t h) { h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29; return h; }
__device__ __forceinline__ uint32_t dag_hash_children(const uint32_t c[8]) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int i = 0; i < 8; i += 2) h = dag_mix_dev(h ^ (uint64_t(c[i]) | (uint64_t(c[i + 1]) << 32))) + uint64_t(i);
    return uint32_t(dag_mix_dev(h) >> 17);
}
__device__ __forceinline__ uint32_t dag_hash_cluster(uint64_t v) { return uint32_t(dag_mix_dev(v ^ 0x9E3779B97F4A7C15ull) >> 17); }
// a stored node [mask, children of the set bits ...] against a candidate's eight child slots (0 = no child: address 0 is never a child)
__device__ __forceinline__ bool dag_node_equals(const uint32_t* __restrict__ raw, uint32_t addr, const uint32_t c[8]) {
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) mask |= c[i] ? 1u << i : 0u;
    if (raw[addr] != mask) return false;
    uint32_t at = addr + 1;
    bool same = true;
#pragma unroll
    for (int i = 0; i < 8; i++) if (c[i]) same = same && raw[at++] == c[i];
    return same;
}
__device__ __forceinline__ void dag_load8(const uint32_t* __restrict__ p, uint32_t c[8]) {
    const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
    c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
}

// ---- u64 block scan (two 32-bit fields per value: both stay below 2^32 over a whole level) -----------------------------------
__device__ __forceinline__ unsigned long long dag_block_scan(unsigned long long v, unsigned long long* total) {   // 256 threads; exclusive
    __shared__ unsigned long long wave_sum_[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(incl, o); if (int(lane) >= o) incl += t; }
    __syncthreads();
    if (lane == 63u) wave_sum_[wave] = incl;
    __syncthreads();
    unsigned long long before = 0;
    for (uint32_t w = 0; w < wave; w++) before += wave_sum_[w];
    *total = wave_sum_[0] + wave_sum_[1] + wave_sum_[2] + wave_sum_[3];
    return before + incl - v;
}
// ---- steps over chunks in sequence order ------------------------------------------------------------------------------------------
// A step over chunks in sequence order: workgroup b takes a CONTIGUOUS run of chunks, so what lies before its first chunk is a plain sum
// of the chunk totals before it (no scan step, no barrier for one).
struct DagRun { uint32_t lo, hi; };
__device__ __forceinline__ DagRun dag_my_chunks(uint32_t chunks) {
    const uint32_t per = (chunks + gridDim.x - 1) / gridDim.x, lo = min(blockIdx.x * per, chunks);
    return DagRun{lo, min(lo + per, chunks)};
}
__device__ __forceinline__ unsigned long long dag_sum_of(const unsigned long long* sums, uint32_t n) {   // every thread of the workgroup calls it
    unsigned long long mine = 0, total;
    for (uint32_t i = threadIdx.x; i < n; i += 256) mine += sums[i];
    dag_block_scan(mine, &total);
    return total;
}

// ---- step 0: the blocks that hold leaves, and the cluster level's candidates -------------------------------------------------
// per block: [valid : 32 | non-empty octets : 32]
constexpr uint32_t kDagStepGrid = 2048;   // workgroups of a chunked step at most (each sums the chunk totals before its run)
constexpr uint32_t kDagBlockChunk = 256;   // blocks per chunk of the two block steps: one block (64 bytes of cluster words) per thread
// (omask[b]: which of block b's eight words hold leaves -- written by the kernels that make the words, so that counting the blocks
// does not read their 64-byte rows again)
__device__ __forceinline__ void dag_step_block_sums(const uint8_t* __restrict__ omask, uint32_t nb, unsigned long long* sums) {
    const DagRun run = dag_my_chunks((nb + kDagBlockChunk - 1) / kDagBlockChunk);
    for (uint32_t c = run.lo; c < run.hi; c++) {
        const uint32_t b = c * kDagBlockChunk + threadIdx.x;
        const uint32_t octets = b < nb ? uint32_t(__popc(uint32_t(omask[b]))) : 0u;
        unsigned long long total;
        dag_block_scan(octets ? (1ull << 32) | octets : 0ull, &total);   // a brick that was claimed but never written has no counterpart in the octree (dag.hpp)
        if (threadIdx.x == 0) sums[c] = total;
    }
}
// The cluster level's sequence of additions (submap.hpp:70-99): for every non-empty octet, in block and octet order, the
// signed-distance word and then the weight word.  saturate == 0 (the reference's own behaviour, Q6): every weight word is all
// ones, so the sequence is [t0, ones, t1, t2, ...] -- the constant is looked up once, as the second candidate, and the other
// additions of it are counted as duplicates.  saturate != 0: [t0, w0, t1, w1, ...].
__device__ __forceinline__ void dag_step_block_apply(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ tsdf, const uint64_t* __restrict__ weight, const uint8_t* __restrict__ omask,
                                                     uint32_t nb, uint32_t saturate, const unsigned long long* sums, uint64_t* vkeys, uint32_t* vblock, uint32_t* oct_before, uint64_t* cand) {
    __shared__ uint32_t first_octet[kDagBlockChunk];
    __shared__ uint8_t mask_of[kDagBlockChunk];
    const DagRun run = dag_my_chunks((nb + kDagBlockChunk - 1) / kDagBlockChunk);
    unsigned long long before = dag_sum_of(sums, run.lo);
    for (uint32_t c = run.lo; c < run.hi; c++) {
        const uint32_t b = c * kDagBlockChunk + threadIdx.x;
        const uint32_t mask = b < nb ? uint32_t(omask[b]) : 0u, octets = uint32_t(__popc(mask));
        const unsigned long long v = octets ? (1ull << 32) | octets : 0ull;
        unsigned long long total;
        const unsigned long long at = before + dag_block_scan(v, &total);
        before += total;
        first_octet[threadIdx.x] = uint32_t(at); mask_of[threadIdx.x] = uint8_t(mask);
        if (v) { const uint32_t rank = uint32_t(at >> 32); vkeys[rank] = keys[b]; vblock[rank] = b; oct_before[rank] = uint32_t(at); }
        __syncthreads();
        // the chunk's words, a word per thread and turn (a block's row of eight is one line): a thread per block read its row in a chain
        uint64_t word[8], wword[8];
        uint32_t at_k[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t w = j * kDagBlockChunk + threadIdx.x, lb = w >> 3, o = w & 7u, m = mask_of[lb];
            const bool on = (m >> o) & 1u;
            at_k[j] = on ? first_octet[lb] + uint32_t(__popc(m & ((1u << o) - 1u))) : 0xffffffffu;
            const size_t from = (size_t(c) * kDagBlockChunk + lb) * 8 + o;
            word[j] = on ? tsdf[from] : 0ull;
            wword[j] = on && saturate ? weight[from] : 0ull;
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t k = at_k[j];
            if (k == 0xffffffffu) continue;
            if (saturate) { cand[2 * size_t(k)] = word[j]; cand[2 * size_t(k) + 1] = wword[j]; }
            else { cand[k ? k + 1 : 0] = word[j]; if (!k) cand[1] = kClusterEmpty; }
        }
        __syncthreads();
    }
}
// candidate index of the k-th non-empty octet's signed-distance word / weight word
__device__ __forceinline__ uint32_t dag_cluster_cand_t(uint32_t k, uint32_t saturate) { return saturate ? 2u * k : (k ? k + 1u : 0u); }
__device__ __forceinline__ uint32_t dag_cluster_cand_w(uint32_t k, uint32_t saturate) { return saturate ? 2u * k + 1u : 1u; }

// Step 0 in three launches: chunk totals | the valid blocks' keys and candidates | how many nodes every level gets from this submap:
// block b > 0 opens a new node at every level below the first digit in which its key parts from its predecessor's (digits 0..18
// name a block: three bits per level, root first -- octree.hpp:45-47).
__global__ void __launch_bounds__(256)
k_dag_block_sums(const uint8_t* __restrict__ omask, uint32_t nb, unsigned long long* __restrict__ sums, uint32_t* __restrict__ hist /* [32]: zeroed for k_dag_parted_histogram */) {
    if (blockIdx.x == 0 && threadIdx.x < 32) hist[threadIdx.x] = 0u;
    dag_step_block_sums(omask, nb, sums);
}
__global__ void __launch_bounds__(256)
k_dag_block_apply(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ tsdf, const uint64_t* __restrict__ weight, const uint8_t* __restrict__ omask, uint32_t nb, uint32_t saturate,
                  const unsigned long long* __restrict__ sums, uint64_t* __restrict__ vkeys, uint32_t* __restrict__ vblock, uint32_t* __restrict__ oct_before,
                  uint64_t* __restrict__ cand, unsigned long long* __restrict__ total /* [valid blocks : 32 | non-empty octets : 32] */) {
    dag_step_block_apply(keys, tsdf, weight, omask, nb, saturate, sums, vkeys, vblock, oct_before, cand);
    if (blockIdx.x == 0) {
        const unsigned long long all = dag_sum_of(sums, (nb + kDagBlockChunk - 1) / kDagBlockChunk);
        if (threadIdx.x == 0) *total = all;
    }
}
__global__ void __launch_bounds__(256)
k_dag_parted_histogram(const uint64_t* __restrict__ vkeys, const unsigned long long* __restrict__ total, uint32_t* __restrict__ hist /* [32], zeroed */) {
    __shared__ uint32_t local[32];
    const uint32_t nbv = uint32_t(*total >> 32);   // (the host does not know it yet: the grid is sized for all blocks)
    if (threadIdx.x < 32) local[threadIdx.x] = 0;
    __syncthreads();
    for (size_t b = size_t(blockIdx.x) * blockDim.x + threadIdx.x + 1; b < nbv; b += size_t(gridDim.x) * blockDim.x) {
        const uint64_t x = vkeys[b] ^ vkeys[b - 1];
        const uint32_t parted = ((62u - (63u - uint32_t(__clzll((long long)x)))) / 3u) & 31u;   // dag.hpp: first depth at which the two paths part
        // (neighbouring blocks part at the deepest levels nearly always: one LDS atomic per value the wave holds, not per lane)
        for (uint64_t left = __ballot(true); left;) {
            const uint32_t d0 = uint32_t(__builtin_amdgcn_readlane(int(parted), int(__ffsll((unsigned long long)left)) - 1));
            const uint64_t peers = __ballot(parted == d0) & left;
            if (parted == d0 && (peers & ((1ull << (threadIdx.x & 63u)) - 1ull)) == 0ull) atomicAdd(&local[d0], uint32_t(__popcll(peers)));
            left &= ~peers;
        }
    }
    __syncthreads();
    if (threadIdx.x < 32 && local[threadIdx.x]) atomicAdd(&hist[threadIdx.x], local[threadIdx.x]);
}

// ---- a level's batch: probe | count | commit -----------------------------------------------------------------------------------
// kNode: candidates are rows of eight child addresses (0 = no child); else 64-bit cluster words.
struct DagLevelRef { uint32_t* table; void* raw; uint32_t mask, capacity; };

template <bool kNode>
__device__ __forceinline__ void dag_step_probe(const DagLevelRef& L, const void* cand_, uint32_t n, uint32_t base, uint32_t* pos) {
    // addresses below `base` are stored values; base + i marks a slot claimed by candidate i of this batch
    uint32_t* table = L.table;
    const uint32_t mask = L.mask;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        uint32_t mine[8];
        uint64_t value = 0;
        uint32_t p;
        if (kNode) { dag_load8(static_cast<const uint32_t*>(cand_) + i * 8, mine); p = dag_hash_children(mine) & mask; }
        else { value = static_cast<const uint64_t*>(cand_)[i]; p = dag_hash_cluster(value) & mask; }
        const uint32_t me = base + uint32_t(i);
        for (;; p = (p + 1u) & mask) {
            uint32_t v = __hip_atomic_load(&table[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == 0u) {
                v = atomicCAS(&table[p], 0u, me);
                if (v == 0u) break;   // claimed
            }
            bool same;
            if (v >= base) {   // claimed in this batch by candidate j: the same value?
                const size_t j = v - base;
                if (kNode) {
                    uint32_t other[8];
                    dag_load8(static_cast<const uint32_t*>(cand_) + j * 8, other);
                    same = true;
#pragma unroll
                    for (int k = 0; k < 8; k++) same = same && other[k] == mine[k];
                } else same = static_cast<const uint64_t*>(cand_)[j] == value;
                if (same && v > me) atomicMin(&table[p], me);   // the first candidate of a value owns its slot
            } else same = kNode ? dag_node_equals(static_cast<const uint32_t*>(L.raw), v, mine) : static_cast<const uint64_t*>(L.raw)[v] == value;
            if (same) break;
        }
        pos[i] = p;
    }
}
// words candidate i adds to the store: 0 unless it owns its slot (the first candidate of a value the store does not hold yet)
template <bool kNode>
__device__ __forceinline__ uint32_t dag_new_words(const uint32_t* table, const uint32_t* pos, const uint32_t* cand, uint32_t base, uint32_t i) {
    if (__hip_atomic_load(&table[pos[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != base + i) return 0u;
    uint32_t words = 1;
    if (kNode) { uint32_t c[8]; dag_load8(cand + size_t(i) * 8, c); for (int k = 0; k < 8; k++) words += c[k] ? 1u : 0u; }
    return words;
}
template <bool kNode>
__device__ __forceinline__ void dag_step_count(const DagLevelRef& L, const uint32_t* pos, const void* cand, uint32_t n, uint32_t base, uint8_t* new_words, unsigned long long* sums) {
    const DagRun run = dag_my_chunks((n + kDagChunk - 1) / kDagChunk);
    for (uint32_t c = run.lo; c < run.hi; c++) {
        unsigned long long mine = 0;   // [new values : 32 | their words : 32]
        for (uint32_t j = 0; j < kDagPer; j++) {
            const uint32_t i = c * kDagChunk + threadIdx.x * kDagPer + j;
            if (i >= n) break;
            const uint32_t w = dag_new_words<kNode>(L.table, pos, static_cast<const uint32_t*>(cand), base, i);
            new_words[i] = uint8_t(w);   // (kept: once values are committed a slot no longer tells who owned it)
            mine += w ? (1ull << 32) | w : 0ull;
        }
        unsigned long long total;
        dag_block_scan(mine, &total);
        if (threadIdx.x == 0) sums[c] = total;
    }
}
// the new values get their addresses behind the store's end in sequence order -- the order of first occurrence, as the reference
// hands them out -- append their words and put the address into their slot; result[i] = 0 for a candidate that was not new
template <bool kNode>
__device__ __forceinline__ void dag_step_commit(const DagLevelRef& L, const uint32_t* pos, const void* cand_, uint32_t n, const uint8_t* new_words, const unsigned long long* sums,
                                                uint32_t base, DagScalars* s, unsigned long long extra_dupes, uint32_t* result) {
    const uint32_t chunks = (n + kDagChunk - 1) / kDagChunk;
    const DagRun run = dag_my_chunks(chunks);
    unsigned long long before = dag_sum_of(sums, run.lo);
    for (uint32_t c = run.lo; c < run.hi; c++) {
        unsigned long long v[kDagPer], mine = 0;
        for (uint32_t j = 0; j < kDagPer; j++) {
            const uint32_t i = c * kDagChunk + threadIdx.x * kDagPer + j;
            v[j] = i < n && new_words[i] ? (1ull << 32) | new_words[i] : 0ull;
            mine += v[j];
        }
        unsigned long long total;
        uint32_t at = base + uint32_t(before + dag_block_scan(mine, &total));   // (low field: words before this candidate among the batch's new values)
        before += total;
        for (uint32_t j = 0; j < kDagPer; j++) {
            const uint32_t i = c * kDagChunk + threadIdx.x * kDagPer + j;
            if (i >= n) break;
            if (!v[j]) { result[i] = 0u; continue; }
            const uint32_t words = uint32_t(v[j]);
            if (at + words <= L.capacity) {   // (the host sizes the store for the worst case: always true)
                if (kNode) {
                    uint32_t ch[8];
                    dag_load8(static_cast<const uint32_t*>(cand_) + size_t(i) * 8, ch);
                    uint32_t* raw = static_cast<uint32_t*>(L.raw);
                    uint32_t mask = 0, w = at + 1;
                    for (int k = 0; k < 8; k++) if (ch[k]) { mask |= 1u << k; raw[w++] = ch[k]; }
                    raw[at] = mask;
                } else static_cast<uint64_t*>(L.raw)[at] = static_cast<const uint64_t*>(cand_)[i];
            }
            L.table[pos[i]] = at;
            result[i] = at;
            at += words;
        }
    }
    if (blockIdx.x == 0) {   // the batch's totals into the level's scalars (nobody reads `size` inside this launch: the count step left the batch's base in batch_base)
        const unsigned long long all = dag_sum_of(sums, chunks);
        if (threadIdx.x == 0) {
            const uint32_t fresh = uint32_t(all >> 32), words = uint32_t(all);
            s->size = base + words;
            s->entries += fresh;
            s->uniques += fresh;
            s->dupes += (unsigned long long)(n - fresh) + extra_dupes;
        }
    }
}
// the address of candidate i's value behind a batch: its own if it was new (commit wrote it), else what its slot holds now.  (Read
// where it is needed instead -- two dependent loads inside the consumers' per-node loops -- the launch saved costs 0.2 ms per cfg5 fold.)
__device__ __forceinline__ void dag_step_resolve(const uint32_t* table, const uint32_t* pos, uint32_t n, uint32_t* result) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        if (!result[i]) result[i] = table[pos[i]];
}

// ---- the node levels' candidates -------------------------------------------------------------------------------------------
// level 19 (one node per block that holds leaves, submap.hpp:70-99): the eight cluster addresses of each tree
__device__ __forceinline__ void dag_step_block_nodes(const uint8_t* __restrict__ omask, const uint32_t* __restrict__ vblock, const uint32_t* __restrict__ oct_before, uint32_t nbv,
                                                     uint32_t saturate, const uint32_t* cluster_addr /* resolved */, uint32_t* children /* [2 nbv][8] */, const uint64_t* __restrict__ vkeys, uint64_t* prefix) {
    for (size_t r = size_t(blockIdx.x) * blockDim.x + threadIdx.x; r < nbv; r += size_t(gridDim.x) * blockDim.x) {
        prefix[r] = vkeys[r] >> 6;   // the block's digits 0..18: its node's Morton prefix
        const uint32_t mask = omask[vblock[r]];
        uint32_t k = oct_before[r];
        uint32_t t[8], w[8];
#pragma unroll
        for (uint32_t o = 0; o < 8; o++) {
            const bool there = (mask >> o) & 1u;
            t[o] = there ? cluster_addr[dag_cluster_cand_t(k, saturate)] : 0u;
            w[o] = there ? cluster_addr[dag_cluster_cand_w(k, saturate)] : 0u;
            k += there ? 1u : 0u;
        }
        uint4* out = reinterpret_cast<uint4*>(children + r * 16);
        out[0] = make_uint4(t[0], t[1], t[2], t[3]); out[1] = make_uint4(t[4], t[5], t[6], t[7]);
        out[2] = make_uint4(w[0], w[1], w[2], w[3]); out[3] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}
// level d from level d + 1: node j of the level below (prefix[j] = its digits 0..d, i.e. its Morton prefix) belongs to the parent
// prefix[j] >> 3 and sits in child slot prefix[j] & 7.  Heads (first child of a parent) per chunk; then every head, at its rank,
// leaves its position and its parent's prefix.  (Letting the head gather its parent's rows as well saves a launch and costs three
// times the two steps' time: a thread then walks up to eight heads of up to eight children in a dependent chain.)
__device__ __forceinline__ bool dag_is_head(const uint64_t* prefix, uint32_t i) { return i == 0 || (prefix[i] >> 3) != (prefix[i - 1] >> 3); }
__device__ __forceinline__ void dag_step_head_sums(const uint64_t* prefix, uint32_t n, unsigned long long* sums) {
    const DagRun run = dag_my_chunks((n + kDagChunk - 1) / kDagChunk);
    for (uint32_t c = run.lo; c < run.hi; c++) {
        unsigned long long mine = 0;
        for (uint32_t j = 0; j < kDagPer; j++) { const uint32_t i = c * kDagChunk + threadIdx.x * kDagPer + j; if (i < n && dag_is_head(prefix, i)) mine++; }
        unsigned long long total;
        dag_block_scan(mine, &total);
        if (threadIdx.x == 0) sums[c] = total;
    }
}
__device__ __forceinline__ void dag_step_heads(const uint64_t* prefix, uint32_t n, const unsigned long long* sums, uint32_t parents, uint32_t* head_pos /* [parents + 1] */,
                                               uint64_t* parent_prefix) {
    const DagRun run = dag_my_chunks((n + kDagChunk - 1) / kDagChunk);
    unsigned long long before = dag_sum_of(sums, run.lo);
    if (blockIdx.x == 0 && threadIdx.x == 0) head_pos[parents] = n;
    for (uint32_t c = run.lo; c < run.hi; c++) {
        unsigned long long mine = 0;
        uint32_t heads = 0;
        for (uint32_t j = 0; j < kDagPer; j++) {
            const uint32_t i = c * kDagChunk + threadIdx.x * kDagPer + j;
            if (i < n && dag_is_head(prefix, i)) { heads |= 1u << j; mine++; }
        }
        unsigned long long total;
        uint32_t at = uint32_t(before + dag_block_scan(mine, &total));
        before += total;
        for (uint32_t j = 0; j < kDagPer; j++) {
            if (!((heads >> j) & 1u)) continue;
            const uint32_t i = c * kDagChunk + threadIdx.x * kDagPer + j;
            if (at < parents) { head_pos[at] = i; parent_prefix[at] = prefix[i] >> 3; }
            at++;
        }
    }
}
// a parent's two rows of child addresses: its children are the nodes [head_pos[p], head_pos[p + 1]) of the level below (at most eight)
__device__ __forceinline__ void dag_step_parents(const uint64_t* prefix, const uint32_t* head_pos, uint32_t parents, const uint32_t* child_addr /* [2 n], resolved */,
                                                 uint32_t* children /* [2 parents][8] */) {
    for (size_t p = size_t(blockIdx.x) * blockDim.x + threadIdx.x; p < parents; p += size_t(gridDim.x) * blockDim.x) {
        uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint32_t first = head_pos[p], end = head_pos[p + 1];
        for (uint32_t j = first; j < end && j < first + 8u; j++) {
            const uint32_t digit = uint32_t(prefix[j]) & 7u;
            const uint2 ab = reinterpret_cast<const uint2*>(child_addr)[j];
#pragma unroll
            for (uint32_t o = 0; o < 8; o++) { t[o] = o == digit ? ab.x : t[o]; w[o] = o == digit ? ab.y : w[o]; }   // (no runtime-indexed registers)
        }
        uint4* out = reinterpret_cast<uint4*>(children + p * 16);
        out[0] = make_uint4(t[0], t[1], t[2], t[3]); out[1] = make_uint4(t[4], t[5], t[6], t[7]);
        out[2] = make_uint4(w[0], w[1], w[2], w[3]); out[3] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

// ---- the steps as launches --------------------------------------------------------------------------------------------------------
// probe (and, beside it, the head count of the level's own nodes for the parents' step of the level above) | count | commit
template <bool kNode>
__global__ void __launch_bounds__(256)
k_dag_probe(DagLevelRef L, const DagScalars* __restrict__ s, const void* __restrict__ cand, uint32_t n, uint32_t* __restrict__ pos,
            const uint64_t* __restrict__ heads_of, uint32_t heads_n, unsigned long long* __restrict__ head_sums) {
    dag_step_probe<kNode>(L, cand, n, s->size, pos);
    if (heads_of) dag_step_head_sums(heads_of, heads_n, head_sums);
}
template <bool kNode>
__global__ void __launch_bounds__(256)
k_dag_count(DagLevelRef L, DagScalars* __restrict__ s, const uint32_t* __restrict__ pos, const void* __restrict__ cand, uint32_t n, uint8_t* __restrict__ new_words,
            unsigned long long* __restrict__ sums) {
    const uint32_t base = s->size;
    if (blockIdx.x == 0 && threadIdx.x == 0) s->batch_base = base;
    dag_step_count<kNode>(L, pos, cand, n, base, new_words, sums);
}
template <bool kNode>
__global__ void __launch_bounds__(256)
k_dag_commit(DagLevelRef L, DagScalars* s, const uint32_t* __restrict__ pos, const void* __restrict__ cand, uint32_t n, const uint8_t* __restrict__ new_words,
             const unsigned long long* __restrict__ sums, unsigned long long extra_dupes, uint32_t* __restrict__ result) {
    dag_step_commit<kNode>(L, pos, cand, n, new_words, sums, s->batch_base, s, extra_dupes, result);
}
__global__ void __launch_bounds__(256)
k_dag_resolve(const uint32_t* __restrict__ table, const uint32_t* __restrict__ pos, uint32_t n, uint32_t* __restrict__ result) { dag_step_resolve(table, pos, n, result); }
__global__ void __launch_bounds__(256)
k_dag_block_nodes(const uint8_t* __restrict__ omask, const uint32_t* __restrict__ vblock, const uint32_t* __restrict__ oct_before, uint32_t nbv, uint32_t saturate,
                  const uint32_t* __restrict__ cluster_addr, uint32_t* __restrict__ children, const uint64_t* __restrict__ vkeys, uint64_t* __restrict__ prefix) {
    dag_step_block_nodes(omask, vblock, oct_before, nbv, saturate, cluster_addr, children, vkeys, prefix);
}
__global__ void __launch_bounds__(256)
k_dag_heads(const uint64_t* __restrict__ prefix, uint32_t n, const unsigned long long* __restrict__ sums, uint32_t parents, uint32_t* __restrict__ head_pos,
            uint64_t* __restrict__ parent_prefix) {
    dag_step_heads(prefix, n, sums, parents, head_pos, parent_prefix);
}
__global__ void __launch_bounds__(256)
k_dag_parents(const uint64_t* __restrict__ prefix, const uint32_t* __restrict__ head_pos, uint32_t parents, const uint32_t* __restrict__ child_addr, uint32_t* __restrict__ children) {
    dag_step_parents(prefix, head_pos, parents, child_addr, children);
}

constexpr uint32_t kDagSmall = 1024;   // nodes of a level below which one workgroup builds everything above it (k_dag_small_levels)
struct DagFoldScratch { uint64_t* prefix; uint32_t* children; uint32_t* pos; uint32_t* addr; uint32_t* head_pos; };

// ---- the upper levels in ONE launch -------------------------------------------------------------------------------------------
// Above the first few levels a submap has a handful of nodes per level (a cfg2 room: 9 409 blocks, 1 500 nodes one level up, 250,
// 50, ... 1), and a level costs four dependent launches in the generic form above.  One workgroup of 1 024 threads runs the same
// steps for every level from `top` down to the root, as long as the level below has at most kDagSmall nodes: candidates in LDS,
// the tables and stores in global memory as before.
struct DagLevelsArg { uint32_t* table[kDagDepth]; uint32_t* raw[kDagDepth]; uint32_t mask[kDagDepth]; uint32_t capacity[kDagDepth]; };

__device__ __forceinline__ uint32_t dag_scan1024(uint32_t v, uint32_t* total, uint32_t* scratch /* 17 words of LDS */) {   // exclusive, 1 024 threads
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (int(lane) >= o) incl += t; }
    __syncthreads();
    if (lane == 63u) scratch[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < 16; w++) { const uint32_t t = scratch[w]; before += w < wave ? t : 0u; all += t; }
    *total = all;
    return before + incl - v;
}

__global__ void __launch_bounds__(1024)
k_dag_small_levels(DagLevelsArg L, DagScalars* __restrict__ scalars, int top /* first level to build */, uint32_t n_below,
                   const uint64_t* __restrict__ prefix_below, const uint32_t* __restrict__ addr_below /* [2 n_below], resolved */, uint32_t* __restrict__ roots_out) {
    __shared__ uint64_t prefix[2][kDagSmall];
    __shared__ uint32_t addr[2][2 * kDagSmall];
    __shared__ uint32_t cand[2 * kDagSmall][8];
    __shared__ uint32_t head_pos[kDagSmall + 1];
    __shared__ uint32_t scratch[17];
    __shared__ uint32_t level_size[kDagDepth];   // the levels' sizes, read once: nobody else adds to these levels while this kernel runs
    const uint32_t t = threadIdx.x;
    uint32_t cur = 0;
    if (t < kDagDepth) level_size[t] = scalars[t].size;
    if (t < n_below) { prefix[0][t] = prefix_below[t]; addr[0][2 * t] = addr_below[2 * t]; addr[0][2 * t + 1] = addr_below[2 * t + 1]; }
    __syncthreads();
    for (int d = top; d >= 0; d--) {
        // parents of the level below
        const bool head = t < n_below && (t == 0 || (prefix[cur][t] >> 3) != (prefix[cur][t - 1] >> 3));
        uint32_t parents;
        const uint32_t rank = dag_scan1024(head ? 1u : 0u, &parents, scratch);
        if (head) { head_pos[rank] = t; prefix[cur ^ 1][rank] = prefix[cur][t] >> 3; }
        if (n_below == 0u) parents = d == 0 ? 1u : 0u;   // an empty octree still yields a childless root pair (submap.hpp:29-44)
        if (t == 0) head_pos[parents] = n_below;
        __syncthreads();
        if (t < parents) {
            uint32_t ct[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const uint32_t first = n_below ? head_pos[t] : 0u, end = n_below ? head_pos[t + 1] : 0u;
            for (uint32_t j = first; j < end && j < first + 8u; j++) {
                const uint32_t digit = uint32_t(prefix[cur][j]) & 7u, a = addr[cur][2 * j], b = addr[cur][2 * j + 1];
#pragma unroll
                for (uint32_t o = 0; o < 8; o++) { ct[o] = o == digit ? a : ct[o]; cw[o] = o == digit ? b : cw[o]; }
            }
#pragma unroll
            for (uint32_t o = 0; o < 8; o++) { cand[2 * t][o] = ct[o]; cand[2 * t + 1][o] = cw[o]; }
        }
        __syncthreads();
        // probe: thread t owns candidates 2t and 2t + 1 (consecutive: sequence order)
        const uint32_t n = 2u * parents;
        uint32_t* table = L.table[d];
        uint32_t* raw = L.raw[d];
        const uint32_t mask = L.mask[d];
        uint32_t pos[2] = {0, 0};
        const uint32_t base = level_size[d];
        for (uint32_t k = 0; k < 2; k++) {
            const uint32_t i = 2 * t + k;
            if (i >= n) break;
            uint32_t mine[8];
#pragma unroll
            for (int o = 0; o < 8; o++) mine[o] = cand[i][o];
            const uint32_t me = base + i;
            uint32_t p = dag_hash_children(mine) & mask;
            for (;; p = (p + 1u) & mask) {
                uint32_t v = __hip_atomic_load(&table[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v == 0u) { v = atomicCAS(&table[p], 0u, me); if (v == 0u) break; }
                bool same;
                if (v >= base) {
                    const uint32_t j = v - base;
                    same = true;
#pragma unroll
                    for (int o = 0; o < 8; o++) same = same && cand[j][o] == mine[o];
                    if (same && v > me) (void)atomicMin(&table[p], me);
                } else same = dag_node_equals(raw, v, mine);
                if (same) break;
            }
            pos[k] = p;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (atomicMin returns nothing: it must have been performed before anybody reads the slot)
        __syncthreads();
        // who owns its slot (new: its words are counted), who found a stored value (its address is what the slot holds), who shares a
        // new value with an earlier candidate of this batch (the owner's address, once the owner has one: from LDS, not from the slot)
        unsigned long long v[2] = {0, 0};
        uint32_t seen[2] = {0, 0};
        for (uint32_t k = 0; k < 2; k++) {
            const uint32_t i = 2 * t + k;
            if (i >= n) break;
            seen[k] = __hip_atomic_load(&table[pos[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen[k] == base + i) {
                uint32_t words = 1;
                for (int o = 0; o < 8; o++) words += cand[i][o] ? 1u : 0u;
                v[k] = (1ull << 32) | words;
            }
        }
        uint32_t total_words, total_new;
        uint32_t at = base + dag_scan1024(uint32_t(v[0]) + uint32_t(v[1]), &total_words, scratch);
        (void)dag_scan1024(uint32_t(v[0] >> 32) + uint32_t(v[1] >> 32), &total_new, scratch);
        if (t == 0) {   // (stores and result-less atomics only: nothing here waits for memory)
            DagScalars* s = scalars + d;
            s->batch_base = base;
            __hip_atomic_store(&s->size, base + total_words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&s->entries, total_new);
            atomicAdd(&s->uniques, (unsigned long long)total_new);
            atomicAdd(&s->dupes, (unsigned long long)(n - total_new));
        }
        for (uint32_t k = 0; k < 2; k++) {
            const uint32_t i = 2 * t + k;
            if (i >= n) break;
            if (!v[k]) { if (seen[k] < base) addr[cur ^ 1][i] = seen[k]; continue; }
            const uint32_t words = uint32_t(v[k]);
            if (at + words <= L.capacity[d]) {
                uint32_t m = 0, w = at + 1;
                for (int o = 0; o < 8; o++) if (cand[i][o]) { m |= 1u << o; raw[w++] = cand[i][o]; }
                raw[at] = m;
            }
            table[pos[k]] = at;   // (nobody reads this level's table again in this launch)
            addr[cur ^ 1][i] = at;
            at += words;
        }
        __syncthreads();
        for (uint32_t k = 0; k < 2; k++) {
            const uint32_t i = 2 * t + k;
            if (i < n && !v[k] && seen[k] >= base) addr[cur ^ 1][i] = addr[cur ^ 1][seen[k] - base];
        }
        __syncthreads();
        cur ^= 1;
        n_below = parents;
    }
    if (t < 2) roots_out[t] = addr[cur][t];
}

// ---- table growth ------------------------------------------------------------------------------------------------------------
template <bool kNode>
__global__ void __launch_bounds__(256)
k_dag_rehash(const uint32_t* __restrict__ old_table, uint32_t old_slots, const void* __restrict__ raw_, uint32_t* __restrict__ table, uint32_t mask) {
    for (size_t s = size_t(blockIdx.x) * blockDim.x + threadIdx.x; s < old_slots; s += size_t(gridDim.x) * blockDim.x) {
        const uint32_t a = old_table[s];
        if (!a) continue;
        uint32_t p;
        if (kNode) {
            const uint32_t* raw = static_cast<const uint32_t*>(raw_);
            uint32_t c[8], at = a + 1;
            const uint32_t m = raw[a];
#pragma unroll
            for (int k = 0; k < 8; k++) c[k] = (m >> k) & 1u ? raw[at++] : 0u;
            p = dag_hash_children(c) & mask;
        } else p = dag_hash_cluster(static_cast<const uint64_t*>(raw_)[a]) & mask;
        while (atomicCAS(&table[p], 0u, a) != 0u) p = (p + 1u) & mask;   // (stored values are distinct: no comparison needed)
    }
}

// ==== host side ===============================================================================================================
struct DeviceDag {
    DagLevelDev level[kDagDepth + 1];          // [0..19] node levels, [20] the cluster level
    DagScalars* d_scalars = nullptr;           // [21]
    DagScalars h_scalars[kDagDepth + 1] = {};  // as of the last fold
    unsigned long long* d_total = nullptr;     // [0]: the blocks' totals
    uint32_t* d_hist = nullptr;                // [32]: nodes opened per depth, then two words for the roots
    uint32_t* h_small = nullptr;               // pinned: what a fold reads back in the middle (block totals, level histogram) and at its end (roots, scalars)
    char* arena = nullptr; size_t arena_cap = 0;
    DevPool pool;                              // chad_options.dag_pool_bytes: the levels' tables and stores are carved out of this reservation (alloc.hpp); what
                                               // does not fit any more comes from hipMalloc as without a pool
    std::vector<void*> outgrown;               // tables and stores OUTSIDE the pool that a level has grown out of: freed at the next sync (hipFree waits for the
                                               // whole device -- the frames in flight too)
    size_t outgrown_bytes = 0;
    uint64_t growths = 0;                      // tables and stores that have been replaced by bigger ones so far
    uint64_t hint_entries[kDagDepth + 1] = {}, hint_words[kDagDepth + 1] = {};   // chad_map_dag_reserve: what the levels are expected to hold in the end -- they are made that size
    bool ready = false;

    // A range of the pool that a level gives up for a bigger one is free for the NEXT allocation at once: what makes that safe is stream
    // order -- whoever gets the range writes it with work enqueued BEHIND the copy / rehash that still reads it.  The levels are touched
    // from two streams (the fold's, and the map's for submaps folded start to finish): the stream of the last retirement is remembered with
    // an event behind it, and an allocation made for work on another stream makes that stream wait for the event first.
    hipStream_t retired_on = nullptr;
    hipEvent_t retired_event = nullptr;
    bool retired_any = false, pool_corrupt = false;   // pool_corrupt: a range was given back that the pool never handed out (reported by the next fold)
    // a table or store for a level, to be used by work on stream `s`: from the pool while it lasts
    hipError_t store_alloc(void** out, size_t bytes, hipStream_t s) {
        if (void* p = pool.take(bytes)) {
            if (retired_any && s != retired_on) { const hipError_t e = hipStreamWaitEvent(s, retired_event, 0); if (e != hipSuccess) { (void)pool.give(p, bytes); return e; } }
            *out = p;
            return hipSuccess;
        }
        return chad_malloc(out, bytes);
    }
    // ... given up for a bigger one; the last work that reads it has been enqueued on stream `s`
    void store_retire(void* p, size_t bytes, hipStream_t s) {
        if (!p) return;
        if (pool.owns(p)) {
            if (!pool.give(p, bytes)) { pool_corrupt = true; return; }
            if (!retired_event && hipEventCreateWithFlags(&retired_event, hipEventDisableTiming) != hipSuccess) { retired_event = nullptr; (void)hipGetLastError(); }
            if (retired_event && hipEventRecord(retired_event, s) == hipSuccess) { retired_on = s; retired_any = true; }
        } else { outgrown.push_back(p); outgrown_bytes += bytes; }
    }
    void store_free(void* p, size_t bytes) {   // (never used by the device, or the device is idle)
        if (!p) return;
        if (pool.owns(p)) { if (!pool.give(p, bytes)) pool_corrupt = true; } else (void)chad_free(p);
    }
    void free_outgrown() {   // (the device is idle: nothing is folding, nothing reads a level)
        for (void* p : outgrown) (void)chad_free(p);
        outgrown.clear(); outgrown_bytes = 0;
    }
    // the DAG thrown away (chad_map_reset): the pool's reservation stays -- a map that is reset and filled again allocates nothing
    void release(bool pool_too = false) {
        for (uint32_t l = 0; l <= kDagDepth; l++) {
            DagLevelDev& L = level[l];
            store_free(L.table, size_t(L.table_slots) * 4); store_free(L.raw, size_t(L.raw_capacity) * (l == kDagDepth ? 8 : 4));
            L = DagLevelDev{};
        }
        free_outgrown();
        pool.reset();
        retired_any = false; pool_corrupt = false;
        if (pool_too) { pool.release(); if (retired_event) (void)hipEventDestroy(retired_event); retired_event = nullptr; }
        if (pool_too) {   // (a reset keeps the fold's scratch and the few scalars: a map that is reset and filled again allocates nothing)
            (void)chad_free(d_scalars); (void)chad_free(d_total); (void)chad_free(d_hist); (void)chad_free(arena);
            if (h_small) (void)hipHostFree(h_small);
            d_scalars = nullptr; d_total = nullptr; d_hist = nullptr; h_small = nullptr; arena = nullptr; arena_cap = 0;
        }
        ready = false;
        for (auto& s : h_scalars) s = DagScalars{};
    }
};

inline dim3 dag_grid(size_t n, uint32_t per = 256, uint32_t cap = 2048) { return dim3(unsigned(std::max<size_t>(1, std::min<size_t>(cap, (n + per - 1) / per)))); }

inline hipError_t dag_device_init(DeviceDag& d, hipStream_t s) {
    if (d.ready) return hipSuccess;
    hipError_t e;
    if (!d.d_scalars && (e = chad_malloc(&d.d_scalars, sizeof(DagScalars) * (kDagDepth + 1))) != hipSuccess) return e;
    if (!d.d_total && (e = chad_malloc(&d.d_total, 8 * 64)) != hipSuccess) return e;
    if (!d.d_hist && (e = chad_malloc(&d.d_hist, 4 * 32)) != hipSuccess) return e;
    if (!d.h_small && (e = hipHostMalloc(&d.h_small, 4096)) != hipSuccess) return e;
    for (uint32_t l = 0; l <= kDagDepth; l++) {
        DagLevelDev& L = d.level[l];
        L.table_slots = 1u << 19; L.raw_capacity = 1u << 19;   // (2 + 2..4 MB per level: a room-sized submap's 75 k cluster candidates fit without growing anything)
        if (d.hint_entries[l] || d.hint_words[l]) {   // chad_map_dag_reserve: the level at its expected final size, from the start (a table at most half full)
            uint64_t slots = L.table_slots;
            while (slots < 2 * (d.hint_entries[l] + d.hint_entries[l] / 16) + 16 && slots < (1ull << 31)) slots <<= 1;
            L.table_slots = uint32_t(slots);
            L.raw_capacity = uint32_t(std::min<uint64_t>(std::max<uint64_t>(L.raw_capacity, d.hint_words[l] + d.hint_words[l] / 16 + 4096), (1ull << 32) - 1));
        }
        if ((e = d.store_alloc(reinterpret_cast<void**>(&L.table), size_t(L.table_slots) * 4, s)) != hipSuccess) return e;
        if ((e = d.store_alloc(&L.raw, size_t(L.raw_capacity) * (l == kDagDepth ? 8 : 4), s)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(L.table, 0, size_t(L.table_slots) * 4, s)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(L.raw, 0, l == kDagDepth ? 8 : 4, s)) != hipSuccess) return e;   // address 0: the reserved word (levels.hpp:52, :118)
        d.h_scalars[l] = DagScalars{};
        d.h_scalars[l].size = 1;
    }
    if ((e = hipMemcpyAsync(d.d_scalars, d.h_scalars, sizeof(d.h_scalars), hipMemcpyHostToDevice, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;   // (h_scalars is the source of an asynchronous copy)
    d.ready = true;
    return hipSuccess;
}

// room in level l for `n` more candidates of at most `words_each` words: the table stays at most half full, the store never overflows
inline hipError_t dag_level_reserve(DeviceDag& d, uint32_t l, size_t n, uint32_t words_each, hipStream_t s) {
    DagLevelDev& L = d.level[l];
    const bool cluster = l == kDagDepth;
    const size_t need_slots = 2 * (size_t(d.h_scalars[l].entries) + n) + 16;
    hipError_t e;
    if (d.pool_corrupt) return hipErrorInvalidValue;   // (a range went back to the pool that it never handed out: its free list is not to be trusted)
    static const bool report = std::getenv("CHAD_DEBUG_DAG_GROWTH") != nullptr;   // which level grows, and why (stderr)
    if (report && (need_slots > L.table_slots || size_t(d.h_scalars[l].size) + n * words_each + 16 > L.raw_capacity))
        std::fprintf(stderr, "chad: DAG level %u grows: %zu candidates x %u words; table %u slots (needs %zu), store %u words (needs %zu); entries %u, size %u\n", l, n, words_each,
                     L.table_slots, need_slots, L.raw_capacity, size_t(d.h_scalars[l].size) + n * words_each + 16, d.h_scalars[l].entries, d.h_scalars[l].size);
    if (need_slots > L.table_slots) {
        // With the DAG's memory reserved up front (DeviceDag::pool) a growth is a rehash on the fold's stream and nothing else, so the table
        // takes what keeps it between a quarter and half full; without a pool every growth is a hipMalloc that stalls the device (0.2 ms and
        // more, box dependent), and the table takes room for three times as much again.  (cfg5's trajectory: 59 GB of levels with the
        // generous policy, about half of it slack.)
        const size_t room = d.pool.base ? 2 : 4;
        size_t slots = L.table_slots;
        while (slots < room * need_slots) slots <<= 1;
        slots = std::min<size_t>(slots, size_t(1) << 31);
        if (need_slots > (size_t(1) << 31)) return hipErrorOutOfMemory;   // (a table of 2^31 slots: a billion values in one level)
        uint32_t* bigger = nullptr;
        if ((e = d.store_alloc(reinterpret_cast<void**>(&bigger), slots * 4, s)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(bigger, 0, slots * 4, s)) != hipSuccess) { d.store_free(bigger, slots * 4); return e; }
        if (cluster) hipLaunchKernelGGL(k_dag_rehash<false>, dag_grid(L.table_slots), dim3(256), 0, s, L.table, L.table_slots, L.raw, bigger, uint32_t(slots - 1));
        else hipLaunchKernelGGL(k_dag_rehash<true>, dag_grid(L.table_slots), dim3(256), 0, s, L.table, L.table_slots, L.raw, bigger, uint32_t(slots - 1));
        // (no wait, no hipFree: what follows on this stream comes behind the rehash, and nothing else reads a level while a fold is in flight)
        d.store_retire(L.table, size_t(L.table_slots) * 4, s);
        d.growths++;
        L.table = bigger; L.table_slots = uint32_t(slots);
    }
    const size_t need_words = size_t(d.h_scalars[l].size) + n * words_each + 16;
    if (need_words > L.raw_capacity) {
        size_t words = L.raw_capacity;
        if (d.pool.base) words = need_words + need_words / 2 + 4096;   // (a copy inside the pool, on the fold's stream: half as much room again)
        else while (words < 2 * need_words) words += words / 2 + 1024;
        if (need_words + n >= (size_t(1) << 32)) return hipErrorOutOfMemory;   // addresses are 32 bits, as in the reference (levels.hpp:50, :117); the batch's pending marks lie above them
        words = std::min<size_t>(words, (size_t(1) << 32) - 1 - n);
        const size_t unit = cluster ? 8 : 4;
        void* bigger = nullptr;
        if ((e = d.store_alloc(&bigger, words * unit, s)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(bigger, L.raw, size_t(d.h_scalars[l].size) * unit, hipMemcpyDeviceToDevice, s)) != hipSuccess) { d.store_free(bigger, words * unit); return e; }
        d.store_retire(L.raw, size_t(L.raw_capacity) * unit, s);
        d.growths++;
        L.raw = bigger; L.raw_capacity = uint32_t(words);
    }
    return hipSuccess;
}

// Fold one submap: keys[nb] ascending block keys, tsdf[8 nb] / weight[8 nb] (null: the reference's all-ones weights) cluster words,
// omask[nb] which of a block's eight words hold leaves (tsdf word != kClusterEmpty), all
// on the device and left alone until the fold is done.  The fold needs numbers on the host once on its way (how many blocks hold
// leaves and how many nodes every level gets: the stores are sized for exactly that) and once at its end (the roots, the levels'
// sizes): two stages, the second enqueued when the first stage's numbers have arrived -- dag_fold_advance(block = false) never
// waits, so a roll-over's fold rides beside the next frames (the insert path polls it); block = true waits for the stage's event.
struct DagFold {
    int stage = 0;   // 0: idle; 1: step 0 enqueued (block totals and level histogram on their way); 2: the levels enqueued (roots on their way)
    const uint64_t* keys = nullptr; const uint64_t* tsdf = nullptr; const uint64_t* weight = nullptr; const uint8_t* omask = nullptr;
    uint32_t nb = 0, saturate = 0, nbv = 0, octets = 0;
    uint64_t* vkeys = nullptr; uint32_t* vblock = nullptr; uint32_t* oct_before = nullptr;
    uint64_t* ccand = nullptr; uint32_t* cpos = nullptr; uint32_t* caddr = nullptr;
    unsigned long long* sums = nullptr; unsigned long long* head_sums = nullptr; uint8_t* new_words = nullptr;
    DagFoldScratch ls[2] = {};
    hipEvent_t ready = nullptr;   // behind the device-to-host copy of the current stage's numbers
};

inline hipError_t dag_fold_begin(DeviceDag& d, DagFold& f, const uint64_t* keys, const uint64_t* tsdf, const uint64_t* weight, const uint8_t* omask, uint32_t nb, hipStream_t s) {
    hipError_t e;
    if ((e = dag_device_init(d, s)) != hipSuccess) return e;
    if (!f.ready && (e = hipEventCreateWithFlags(&f.ready, hipEventDisableTiming)) != hipSuccess) return e;
    f.keys = keys; f.tsdf = tsdf; f.weight = weight; f.omask = omask; f.nb = nb; f.saturate = weight ? 1u : 0u; f.nbv = f.octets = 0;
    const size_t cc_ub = f.saturate ? size_t(16) * nb : size_t(8) * nb + 2;       // cluster candidates at most
    const size_t chunks_ub = std::max<size_t>((std::max<size_t>(cc_ub, 2 * size_t(nb) + 2) + kDagChunk - 1) / kDagChunk, (size_t(nb) + kDagBlockChunk - 1) / kDagBlockChunk) + 1;
    const size_t head_chunks_ub = (size_t(nb) + kDagChunk) / kDagChunk + 1;
    auto padded = [](size_t bytes) { return (bytes + 255) & ~size_t(255); };
    const size_t nbp = size_t(nb) + 1;
    const size_t need = padded(nbp * 8) + 2 * padded(nbp * 4) + padded(cc_ub * 8) + 2 * padded(cc_ub * 4) + padded(chunks_ub * 8) + padded(head_chunks_ub * 8) +
                        padded(std::max<size_t>(cc_ub, 2 * nbp)) + 2 * (padded(nbp * 8) + padded(nbp * 64) + 2 * padded(nbp * 8) + padded((nbp + 1) * 4));
    if (need > d.arena_cap) {
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        (void)chad_free(d.arena); d.arena = nullptr; d.arena_cap = 0;
        if ((e = chad_malloc(&d.arena, need + need / 2)) != hipSuccess) return e;   // (grow-only, with room: submaps of a trajectory differ by tens of per cent)
        d.arena_cap = need + need / 2;
    }
    size_t used = 0;
    auto take = [&](size_t bytes) { void* p = d.arena + used; used += padded(bytes); return p; };
    f.vkeys = static_cast<uint64_t*>(take(nbp * 8));
    f.vblock = static_cast<uint32_t*>(take(nbp * 4));
    f.oct_before = static_cast<uint32_t*>(take(nbp * 4));
    f.ccand = static_cast<uint64_t*>(take(cc_ub * 8));
    f.cpos = static_cast<uint32_t*>(take(cc_ub * 4));
    f.caddr = static_cast<uint32_t*>(take(cc_ub * 4));
    f.sums = static_cast<unsigned long long*>(take(chunks_ub * 8));
    f.head_sums = static_cast<unsigned long long*>(take(head_chunks_ub * 8));
    f.new_words = static_cast<uint8_t*>(take(std::max<size_t>(cc_ub, 2 * nbp)));
    for (auto& x : f.ls) {
        x.prefix = static_cast<uint64_t*>(take(nbp * 8)); x.children = static_cast<uint32_t*>(take(nbp * 64));
        x.pos = static_cast<uint32_t*>(take(nbp * 8)); x.addr = static_cast<uint32_t*>(take(nbp * 8)); x.head_pos = static_cast<uint32_t*>(take((nbp + 1) * 4));
    }
    // step 0: the blocks that hold leaves, the cluster candidates, the nodes every level gets
    *reinterpret_cast<volatile unsigned long long*>(d.h_small) = 0ull;
    if (nb) {
        const dim3 bgrid(std::min((nb + kDagBlockChunk - 1) / kDagBlockChunk, kDagStepGrid));
        hipLaunchKernelGGL(k_dag_block_sums, bgrid, dim3(256), 0, s, omask, nb, f.sums, d.d_hist);
        hipLaunchKernelGGL(k_dag_block_apply, bgrid, dim3(256), 0, s, keys, tsdf, weight, omask, nb, f.saturate, f.sums, f.vkeys, f.vblock, f.oct_before, f.ccand, d.d_total);
        hipLaunchKernelGGL(k_dag_parted_histogram, dag_grid(nb, 256, 1024), dim3(256), 0, s, f.vkeys, d.d_total, d.d_hist);
        if ((e = hipMemcpyAsync(d.h_small, d.d_total, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(d.h_small + 16, d.d_hist, 128, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipEventRecord(f.ready, s)) != hipSuccess) return e;
    f.stage = 1;
    return hipSuccess;
}

// One stage further if its numbers are there (block: wait for them).  *done: the fold is complete, roots valid, d.h_scalars up to date.
inline hipError_t dag_fold_advance(DeviceDag& d, DagFold& f, hipStream_t s, bool block, bool* done, uint32_t* root_tsdf, uint32_t* root_weight) {
    hipError_t e;
    *done = false;
    if (f.stage == 0) return hipSuccess;
    if (block) { if ((e = hipEventSynchronize(f.ready)) != hipSuccess) return e; }
    else if (hipEventQuery(f.ready) != hipSuccess) { (void)hipGetLastError(); return hipSuccess; }
    uint32_t* d_roots = d.d_hist + 24;   // (two spare words behind the histogram's twenty)
    if (f.stage == 1) {
        const unsigned long long t = *reinterpret_cast<volatile unsigned long long*>(d.h_small);
        const uint32_t nbv = f.nbv = uint32_t(t >> 32);
        f.octets = uint32_t(t);
        // every store sized for what this submap can add at most: its candidates are counted (n_d = 1 + #{blocks that part from their predecessor at a depth below d})
        const uint32_t ncc = f.octets ? (f.saturate ? 2u * f.octets : f.octets + 1u) : 0u;
        if ((e = dag_level_reserve(d, kDagDepth, ncc, 1, s)) != hipSuccess) return e;
        uint32_t level_nodes[kDagDepth] = {}, running = 1;
        for (uint32_t dd = 0; dd < kDagDepth; dd++) {
            level_nodes[dd] = nbv ? running : 0u;
            running += nbv ? d.h_small[16 + dd] : 0u;
            const uint32_t nodes = nbv ? level_nodes[dd] : (dd == 0 ? 1u : 0u);   // (an empty octree still yields a childless root pair)
            if ((e = dag_level_reserve(d, dd, 2 * size_t(nodes), 9, s)) != hipSuccess) return e;
        }
        auto ref = [&](uint32_t l) { return DagLevelRef{d.level[l].table, d.level[l].raw, d.level[l].table_slots - 1, d.level[l].raw_capacity}; };
        // probe | count | commit | resolve of level l over n candidates; beside the probe, the heads among the level's own nodes for the level above
        auto batch = [&](uint32_t l, const void* cand, uint32_t n, uint32_t* pos, uint32_t* result, unsigned long long extra_dupes, const uint64_t* heads_of, uint32_t heads_n) {
            // (the cluster level's and level 19's batches of a small submap probe with an eighth of the threads, eight candidates each: with
            // every candidate of a room-sized submap in flight at once, the thousands that hold the same few values -- empty-but-one
            // clusters, the all-ones weight nodes -- queue on the same slots: 61 -> 37 and 49 -> 24 us for a cfg2 frame.  Levels above
            // hold few duplicates and lose by it.)
            const dim3 flat = dag_grid(n, l + 1 >= kDagDepth && n < (1u << 17) ? 2048u : 256u), chunked(std::max(1u, std::min((n + kDagChunk - 1) / kDagChunk, kDagStepGrid)));
            DagScalars* sc = d.d_scalars + l;
            if (l == kDagDepth) {
                hipLaunchKernelGGL(k_dag_probe<false>, flat, dim3(256), 0, s, ref(l), sc, cand, n, pos, heads_of, heads_n, f.head_sums);
                hipLaunchKernelGGL(k_dag_count<false>, chunked, dim3(256), 0, s, ref(l), sc, pos, cand, n, f.new_words, f.sums);
                hipLaunchKernelGGL(k_dag_commit<false>, chunked, dim3(256), 0, s, ref(l), sc, pos, cand, n, f.new_words, f.sums, extra_dupes, result);
            } else {
                hipLaunchKernelGGL(k_dag_probe<true>, flat, dim3(256), 0, s, ref(l), sc, cand, n, pos, heads_of, heads_n, f.head_sums);
                hipLaunchKernelGGL(k_dag_count<true>, chunked, dim3(256), 0, s, ref(l), sc, pos, cand, n, f.new_words, f.sums);
                hipLaunchKernelGGL(k_dag_commit<true>, chunked, dim3(256), 0, s, ref(l), sc, pos, cand, n, f.new_words, f.sums, extra_dupes, result);
            }
            hipLaunchKernelGGL(k_dag_resolve, flat, dim3(256), 0, s, static_cast<const uint32_t*>(d.level[l].table), static_cast<const uint32_t*>(pos), n, result);
        };
        if (ncc) batch(kDagDepth, f.ccand, ncc, f.cpos, f.caddr, f.saturate ? 0ull : (unsigned long long)(f.octets - 1u), nullptr, 0u);
        // level 19 from the blocks, then upwards until one workgroup can finish
        uint32_t cur = 0, n_here = nbv;
        int d_at = int(kDagDepth) - 1;
        for (; nbv; d_at--) {
            const DagFoldScratch& here = f.ls[cur];
            const DagFoldScratch& below = f.ls[cur ^ 1];
            if (d_at == int(kDagDepth) - 1)
                hipLaunchKernelGGL(k_dag_block_nodes, dag_grid(nbv), dim3(256), 0, s, f.omask, static_cast<const uint32_t*>(f.vblock), f.oct_before, nbv, f.saturate,
                                   static_cast<const uint32_t*>(f.caddr), here.children, f.vkeys, here.prefix);
            else {
                const uint32_t n_below = level_nodes[d_at + 1];
                hipLaunchKernelGGL(k_dag_heads, dim3(std::max(1u, std::min((n_below + kDagChunk - 1) / kDagChunk, kDagStepGrid))), dim3(256), 0, s, below.prefix, n_below, f.head_sums, n_here,
                                   here.head_pos, here.prefix);
                hipLaunchKernelGGL(k_dag_parents, dag_grid(n_here), dim3(256), 0, s, below.prefix, static_cast<const uint32_t*>(here.head_pos), n_here, static_cast<const uint32_t*>(below.addr), here.children);
            }
            const bool more = n_here > kDagSmall;   // (level 0 has one node: the loop ends)
            batch(uint32_t(d_at), here.children, 2u * n_here, here.pos, here.addr, 0ull, more ? here.prefix : nullptr, n_here);
            if (!more) break;
            n_here = level_nodes[d_at - 1];
            cur ^= 1u;
        }
        // ... which builds level d_at - 1 and everything above it from the n_here nodes of level d_at (an empty octree: its childless root pair)
        DagLevelsArg small{};
        for (uint32_t l = 0; l < kDagDepth; l++) { small.table[l] = d.level[l].table; small.raw[l] = static_cast<uint32_t*>(d.level[l].raw); small.mask[l] = d.level[l].table_slots - 1; small.capacity[l] = d.level[l].raw_capacity; }
        const int top = nbv ? d_at - 1 : int(kDagDepth) - 2;
        hipLaunchKernelGGL(k_dag_small_levels, dim3(1), dim3(1024), 0, s, small, d.d_scalars, top, nbv ? n_here : 0u, static_cast<const uint64_t*>(f.ls[cur].prefix),
                           static_cast<const uint32_t*>(f.ls[cur].addr), d_roots);
        if ((e = hipMemcpyAsync(d.h_small + 64, d_roots, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(d.h_small + 128, d.d_scalars, sizeof(DagScalars) * (kDagDepth + 1), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipEventRecord(f.ready, s)) != hipSuccess) return e;
        f.stage = 2;
        return hipSuccess;
    }
    f.stage = 0;
    *root_tsdf = d.h_small[64]; *root_weight = d.h_small[65];
    DagScalars before[kDagDepth + 1];
    std::memcpy(before, d.h_scalars, sizeof(before));
    std::memcpy(d.h_scalars, d.h_small + 128, sizeof(d.h_scalars));
    *done = true;
    // Growth off the next fold's path: a level that would not take this submap's additions once more grows NOW, behind this fold on its
    // stream and while the frames of the next submap run (found in the middle of a fold -- between its two stages -- an allocation and
    // a rehash held the fold up for 0.2-0.5 ms, in most folds of a map's first hundreds of frames).
    for (uint32_t l = 0; l <= kDagDepth; l++) {
        const size_t entries = d.h_scalars[l].entries - before[l].entries, words = d.h_scalars[l].size - before[l].size;
        if (!entries) continue;
        if (d.hint_entries[l] || d.hint_words[l]) continue;   // (a level made its expected final size needs no room "for one more submap like the last": if the hint was too small it grows when a fold really needs it)
        if ((e = dag_level_reserve(d, l, entries + entries / 2, uint32_t((words + entries - 1) / entries) + 1u, s)) != hipSuccess) return e;
    }
    return hipSuccess;
}

// ... start to finish
inline hipError_t dag_device_add_submap(DeviceDag& d, DagFold& f, const uint64_t* keys, const uint64_t* tsdf, const uint64_t* weight, const uint8_t* omask, uint32_t nb, hipStream_t s,
                                        uint32_t* root_tsdf, uint32_t* root_weight) {
    hipError_t e = dag_fold_begin(d, f, keys, tsdf, weight, omask, nb, s);
    for (bool done = false; e == hipSuccess && !done;) e = dag_fold_advance(d, f, s, true, &done, root_tsdf, root_weight);
    return e;
}

// The host's mirror of the stores (what dag_walk / DagCursor / chad_map_dag_level_copy read): the words added since the last call.
inline hipError_t dag_device_mirror(DeviceDag& d, Dag& host, hipStream_t s) {
    if (!d.ready) return hipSuccess;
    hipError_t e;
    for (uint32_t l = 0; l <= kDagDepth; l++) {
        const size_t size = d.h_scalars[l].size;
        if (l == kDagDepth) {
            const size_t have = host.clusters.raw.size();
            if (have < size) { host.clusters.raw.resize(size); if ((e = hipMemcpyAsync(host.clusters.raw.data() + have, static_cast<uint64_t*>(d.level[l].raw) + have, (size - have) * 8, hipMemcpyDeviceToHost, s)) != hipSuccess) return e; }
            host.clusters.uniques = uint32_t(d.h_scalars[l].uniques); host.clusters.dupes = uint32_t(d.h_scalars[l].dupes);
        } else {
            const size_t have = host.nodes[l].raw.size();
            if (have < size) { host.nodes[l].raw.resize(size); if ((e = hipMemcpyAsync(host.nodes[l].raw.data() + have, static_cast<uint32_t*>(d.level[l].raw) + have, (size - have) * 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return e; }
            host.nodes[l].uniques = uint32_t(d.h_scalars[l].uniques); host.nodes[l].dupes = uint32_t(d.h_scalars[l].dupes);
        }
    }
    return hipStreamSynchronize(s);
}

}  // namespace chadhip


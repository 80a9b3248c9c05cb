// radix_sort.hpp -- stable LSD radix sort of (u64 key, u32 payload) pairs for gfx950, ascending ("onesweep").
//
// Replaces std::sort in sort_morton_vector (morton.hpp:81-102).  Stable, so equal keys keep their input order
// (the deterministic tie order D3 of the oracle).  8-bit digits.  Per sort:
//   k_radix_histograms  one read of the keys: digit histograms of ALL passes at once (LDS atomics), one partial
//                       table per workgroup (no global atomics);
//   k_radix_bases       sums the partials -> keys per digit and pass (the pass kernel scans those 256 values itself);
//   k_radix_onesweep    ONE kernel per pass: a workgroup ranks its tile (wave64 ballot matching inside a wave,
//                       per-wave digit counters in LDS across waves), learns how many keys of each digit precede
//                       its tile from the counts its predecessors publish (32-bit {flag, count} words, so no separate
//                       payload needs release/acquire ordering; two levels: tiles, and groups of 32 tiles), reorders the
//                       tile's keys in LDS and writes runs of equal digits to consecutive addresses.
// The number of passes is data dependent (key width of the frame) and lives in device memory: the host enqueues
// the maximum and surplus passes return at once, so no host round trip sits between the stages of a frame.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>


namespace chadhip {

constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kMaxPasses = 64 / kRadixBits;
// The frame sort may also run with 9-bit digits (512 bins, two per thread of the pass kernel): a key of 17 or 18 bits -- the
// 30 m corridor of cfg3 -- then takes two passes instead of three (25..27 bits: three instead of four).  Chosen per frame on
// the host from the previous frame's key width (radix_digit_bits); buffers are sized for the wider digit.
constexpr int kWideBits = 9;
constexpr int kRadixMax = 1 << kWideBits;
inline uint32_t radix_digit_bits(uint32_t key_bits) { return (key_bits + kWideBits - 1) / kWideBits < (key_bits + kRadixBits - 1) / kRadixBits ? kWideBits : kRadixBits; }
constexpr uint32_t kBaseStripes = 8;  // the frame's keys kernel adds its digit counts into one of this many copies of the totals (atomics on one address serialise)
constexpr int kSortThreads = 256;                       // 4 waves
constexpr int kSortItems = 8;                           // keys per thread
constexpr int kSortTile = kSortThreads * kSortItems;    // 2048 keys per workgroup (1024-key tiles: twice the look-back chain, 11 % slower)
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kLookback = 8;                            // look-back rows read per round trip (independent loads)
constexpr int kHistBlocks = 256;                        // workgroups of the histogram kernel (one partial table each)
constexpr int kMaxHistBlocks = 512;                     // partial tables the buffers have room for

constexpr uint32_t kStatusPublished = 1u << 30;         // flag of a look-back word: the count in its low 30 bits is valid
constexpr uint32_t kStatusValue = (1u << 30) - 1;
constexpr uint32_t kGroupTiles = 32;                    // tiles per group of the two-level look-back
constexpr uint32_t kSpinLimit = 1u << 20;               // polls of one thread for its predecessors before the tile gives up (seconds): a broken
                                                        // dispatch-order assumption then reports kSortFaultStalled instead of hanging the GPU
constexpr uint32_t kSortFaultStalled = 2u;              // bit set in RadixSortSignals::fault (bit 0 is the map's capacity overflow)

// Inclusive sum over the 64 lanes of a wave on the VALU alone: four shifts inside the 16-lane rows (a lane shifted in from outside its row
// reads 0), then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31 into rows 2 and 3 (row_bcast:15 / :31).  Six DPP moves and six
// adds instead of six ds_bpermute round trips with a compare and a select each.  Must be executed by all 64 lanes.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, true));    // row_shr:1
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, true));    // row_shr:2
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, true));    // row_shr:4
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, true));    // row_shr:8
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1 and 3
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2 and 3
    return v;
}

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Element (wave w, step j, lane l) of tile t sits at t*kSortTile + w*(64*kSortItems) + j*64 + l: consecutive
// lanes read consecutive keys, and the tile order (w, j, l) is the input order, which keeps the sort stable.
__device__ __forceinline__ size_t tile_index(uint32_t tile, uint32_t wave, uint32_t j, uint32_t lane) {
    return size_t(tile) * kSortTile + wave * (64 * kSortItems) + j * 64 + lane;
}

// What travels with a key.  IndexPayload: the element's input position (4 bytes).  PointPayload: the point itself and its
// input position as a float4 (x, y, z, index bits) -- 12 more bytes per element and pass, all of them coalesced, instead of
// one scattered 12-byte read per point after the sort (1 M points = 128 MB of 128-byte lines, the most expensive
// memory traffic of the frame); the last pass's output IS the frame's point array in processing order.
struct IndexPayload {
    using T = uint32_t;
    static constexpr bool kStaged = true;         // the payload is reordered through LDS together with its key
    static __device__ __forceinline__ T first(const void*, size_t i) { return uint32_t(i); }
    static __device__ __forceinline__ T none() { return 0u; }
};
struct PointPayload {
    using T = uint4;
    // only the keys go through LDS (20 KB).  Staging the 16-byte payload too costs 32 KB more per workgroup, and three frames'
    // passes then queue for LDS instead of overlapping
    static constexpr bool kStaged = false;        // the payload goes from registers straight to its slot (16 bytes per lane, runs of equal digits adjacent)
    static __device__ __forceinline__ T first(const void* xyz, size_t i) {
        const float* p = static_cast<const float*>(xyz) + 3 * i;
        return make_uint4(__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]), uint32_t(i));
    }
    static __device__ __forceinline__ T none() { return make_uint4(0u, 0u, 0u, 0u); }
};

struct RadixSortBuffers {
    uint64_t* keys[2];
    uint32_t* idx[2];
    uint4* points[2];       // PointPayload ping-pong buffers (null for sorters that only carry indices)
    uint32_t* hist_part;    // [kHistBlocks][kMaxPasses][256] partial digit histograms
    uint32_t* digit_base;   // [kBaseStripes][kMaxPasses][256] keys per digit (totals, the sum over the stripes; every pass workgroup scans them)
    uint32_t* tile_status;  // [kMaxPasses][tiles + groups][256] look-back words, zeroed before every sort
    uint32_t* tile_ticket;  // [kMaxPasses] next tile to hand out, zeroed with tile_status (stored right behind it)
    uint32_t tiles_capacity;
    // Workgroups of the pass kernel that are certainly resident at the same time WHILE EVERYTHING ELSE THAT MAY SHARE THE CHIP
    // RUNS (the host divides the kernel's real occupancy x CU count by the number of sorts in flight): grids up to this size
    // may take their tile from blockIdx and may end in grid barriers; larger ones take tickets.  0 = unknown: always tickets.
    uint32_t resident;
    // ... and of the kernel that FOLLOWS a frame's sort and hosts its catch-all (k_frame_gather, radix_rest_points): that kernel's own
    // occupancy share, so that the catch-all's participants are resident inside it too.  0 = no catch-all: every pass gets its launch.
    uint32_t rest_limit;
};
// Per-sort side channels (all optional).
struct RadixSortSignals {
    uint32_t* fault = nullptr;      // device word: kSortFaultStalled is OR-ed in when a tile gave up waiting (the output is then incomplete)
    uint32_t* consumed = nullptr;   // host-mapped word: every launch after the first stores `seq` here -- pass 0, the only reader of the
    uint32_t seq = 0;               // caller's raw points (PointPayload::first), has completed by then (stream order)
    hipEvent_t* pass_events = nullptr;  // kernel timing: launch p takes its dispatch time stamps into pass_events[2p], [2p + 1]
    uint32_t* launches_out = nullptr;   // (host) ... and the number of pass launches is stored here
    // Passes 5 and 6 (key bits 40..55) are skipped when, over this sort's keys, both digits are FUNCTIONS of the last pass's digit
    // (bits 56..63): keys that agree in the top digit then agree in those two, and sorting by the top digit orders them as well.
    // Block keys are such keys: coordinates are stored with an offset of 2^20 (morton.hpp), so around the origin the upper bits of
    // an axis are 1000..0 or 0111..1 -- all of them change with the sign, none varies on its own -- and far from it they are
    // constant.  k_radix_histograms checks the dependence (dep: [2][256] first digit seen per top digit, stored + 1, then one word
    // that is non-zero when two keys disagree; all zero before the sort, zeroed again by the last pass); two passes, so the result
    // stays in the buffer it would have been in.  Null: every pass runs.
    uint32_t* dep = nullptr;
};
constexpr uint32_t kDepWords = 2 * 256 + 1;
constexpr uint32_t kDepFirstPass = 5, kDepTopPass = 7;   // 8-bit digits: passes 5 and 6 against pass 7

inline uint32_t radix_tiles(size_t n) { return uint32_t((n + kSortTile - 1) / kSortTile); }
__host__ __device__ inline uint32_t radix_groups(uint32_t tiles) { return (tiles + kGroupTiles - 1) / kGroupTiles; }
// look-back words of all passes ([pass][tile rows, then group rows][256]) and 64 ticket / barrier words behind them
__host__ __device__ inline size_t radix_status_words(uint32_t tiles, uint32_t radix = kRadixMax) { return size_t(kMaxPasses) * (tiles + radix_groups(tiles)) * radix + 64; }

// pass_limit (device word, may be null): passes >= *pass_limit are skipped by every kernel.
// Also clears the look-back table of the coming passes (saves a memset launch).
__global__ void __launch_bounds__(kSortThreads)
k_radix_histograms(const uint64_t* __restrict__ keys, uint32_t n, uint32_t passes, const uint32_t* __restrict__ pass_limit,
                   uint32_t* __restrict__ hist_part, uint32_t* __restrict__ tile_status, uint32_t status_words, uint32_t* __restrict__ dep) {
    __shared__ uint32_t hist[kMaxPasses][kRadix];
    __shared__ uint32_t seen[2][kRadix];   // dep: digit of pass 5 / 6 first seen with each top digit, + 1
    __shared__ uint32_t disagree;
    if (dep) { for (uint32_t k = threadIdx.x; k < 2 * kRadix; k += kSortThreads) (&seen[0][0])[k] = 0u; if (threadIdx.x == 0) disagree = 0u; }
    const uint32_t active = pass_limit ? min(passes, *pass_limit) : passes;
    for (size_t k = size_t(blockIdx.x) * kSortThreads + threadIdx.x; k < status_words; k += size_t(gridDim.x) * kSortThreads) tile_status[k] = 0u;
    for (uint32_t k = threadIdx.x; k < kMaxPasses * kRadix; k += kSortThreads) (&hist[0][0])[k] = 0;
    __syncthreads();
    // four keys per thread and turn: their loads are in flight together (one at a time, a workgroup's turns were a chain of round trips)
    const size_t stride = size_t(gridDim.x) * kSortThreads;
    for (size_t i0 = size_t(blockIdx.x) * kSortThreads + threadIdx.x; i0 < n; i0 += 4 * stride) {
        uint64_t key4[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) key4[j] = i0 + j * stride < n ? keys[i0 + j * stride] : 0ull;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            if (i0 + j * stride >= n) break;
            const uint64_t key = key4[j];
            if (dep) {   // (one lane speaks for a wave that agrees on all three digits: the usual case)
                const uint32_t top = uint32_t(key >> (kDepTopPass * kRadixBits)) & (kRadix - 1), both = uint32_t(key >> (kDepFirstPass * kRadixBits)) & 0xffffu;
                const uint32_t word = (top << 16) | both, word0 = uint32_t(__builtin_amdgcn_readfirstlane(int(word)));
                const bool agreed = __ballot(word != word0) == 0ull;
                if (!agreed || (__ballot(true) & ((1ull << (threadIdx.x & 63u)) - 1ull)) == 0ull)
                    for (uint32_t q = 0; q < 2; q++) {
                        const uint32_t mine = ((both >> (8u * q)) & 0xffu) + 1u, was = atomicCAS(&seen[q][top], 0u, mine);
                        if (was != 0u && was != mine) disagree = 1u;
                    }
            }
            // (a digit the whole wave agrees on -- the upper digits of block keys: a submap spans a few values of them -- is counted with
            // one atomic instead of sixty-four on the same LDS word)
            const uint64_t here = __ballot(true);
            for (uint32_t p = 0; p < active; p++) {
                const uint32_t d = uint32_t(key >> (p * kRadixBits)) & (kRadix - 1), d0 = uint32_t(__builtin_amdgcn_readfirstlane(int(d)));
                if (__ballot(d != d0) == 0ull) { if ((here & ((1ull << (threadIdx.x & 63u)) - 1ull)) == 0ull) atomicAdd(&hist[p][d0], uint32_t(__popcll(here))); }
                else atomicAdd(&hist[p][d], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t p = 0; p < active; p++) hist_part[(size_t(blockIdx.x) * kMaxPasses + p) * kRadix + threadIdx.x] = hist[p][threadIdx.x];
    if (dep) {   // this workgroup's view against everybody else's
        bool bad = disagree != 0u;
        for (uint32_t q = 0; q < 2; q++) {
            const uint32_t mine = seen[q][threadIdx.x];
            if (mine) { const uint32_t was = atomicCAS(&dep[q * kRadix + threadIdx.x], 0u, mine); bad = bad || (was != 0u && was != mine); }
        }
        if (bad) atomicOr(&dep[2 * kRadix], 1u);
    }
}

// digit_base[p][d] = number of keys whose digit of pass p is d: the sum of the partial tables.  Four workgroups per pass
// (64 digits each, sixteen slices of the partial tables per digit, 1024 threads) -- one workgroup per pass took 7.7 us.
// The exclusive scan over the digits is done by every workgroup of the pass kernel itself (256 values).
__global__ void __launch_bounds__(4 * kRadix)
k_radix_bases(const uint32_t* __restrict__ hist_part, uint32_t hist_blocks, uint32_t passes, const uint32_t* __restrict__ pass_limit,
              uint32_t* __restrict__ digit_base) {
    const uint32_t p = blockIdx.x >> 2, quarter = blockIdx.x & 3u;
    const uint32_t active = pass_limit ? min(passes, *pass_limit) : passes;
    if (p >= active) return;
    __shared__ uint32_t slice_sum[16][64];
    const uint32_t d = quarter * 64u + (threadIdx.x & 63u), slice = threadIdx.x >> 6;
    uint32_t count = 0;
    const uint32_t per = (hist_blocks + 15) / 16;
    const uint32_t b0 = slice * per, b1 = min(hist_blocks, b0 + per);
#pragma unroll 8
    for (uint32_t b = b0; b < b1; b++) count += hist_part[(size_t(b) * kMaxPasses + p) * kRadix + d];
    slice_sum[slice][threadIdx.x & 63u] = count;
    __syncthreads();
    if (slice != 0) return;
    count = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) count += slice_sum[k][threadIdx.x & 63u];
    digit_base[p * kRadix + d] = count;
}

// One pass for the tile of this workgroup.  pay_in may be null: the payload is then made from the element's position
// (and `first_src`, the raw points, for PointPayload) -- the first pass.  kBits: digit width (8, or 9 with two digits per thread).
// Returns false -- having done nothing -- when the ticket it drew lies beyond the last tile (a workgroup of the catch-all launch
// that goes round for more, radix_rest_points).
template <class P, int kBits>
__device__ __forceinline__ bool
radix_onesweep_tile(const uint64_t* __restrict__ keys_in, const typename P::T* __restrict__ pay_in,
                    uint64_t* __restrict__ keys_out, typename P::T* __restrict__ pay_out,
                    uint32_t n, uint32_t pass, uint32_t tiles, const uint32_t* __restrict__ digit_base, uint32_t base_stripes,
                    uint32_t* __restrict__ tile_status, uint32_t* __restrict__ tile_ticket, const void* __restrict__ first_src, bool use_ticket,
                    uint32_t* __restrict__ fault) {
    constexpr uint32_t kR = 1u << kBits;                  // bins
    constexpr uint32_t kPer = kR / kSortThreads;          // digits per thread: thread t owns t, t + 256, ...
    static_assert(kPer >= 1 && kPer * kSortThreads == kR, "whole digits per thread");
    // the whole input's count of this thread's digits (asked for first: needed only after the tile has been ranked)
    // (all stripes loaded back to back: a loop over base_stripes waited for each load in turn, 2.5 us per pass)
    uint32_t total[kPer];
#pragma unroll
    for (uint32_t q = 0; q < kPer; q++) {
        uint32_t stripe_total[kBaseStripes];
#pragma unroll
        for (uint32_t s = 0; s < kBaseStripes; s++) stripe_total[s] = digit_base[((s < base_stripes ? s : 0u) * kMaxPasses + pass) * kR + threadIdx.x + q * kSortThreads];
        total[q] = 0;
#pragma unroll
        for (uint32_t s = 0; s < kBaseStripes; s++) total[q] += s < base_stripes ? stripe_total[s] : 0u;
    }
    __shared__ uint32_t wave_count[kSortWaves][kR];      // per-wave digit counts, then the wave's base inside the tile
    __shared__ int offset[kR];                           // global slot of local sorted position s is offset[digit] + s
    __shared__ uint32_t wave_total[kSortWaves], wave_total_all[kSortWaves];
    __shared__ uint32_t my_tile, tile_fault;
    __shared__ uint64_t sorted_key[kSortTile];
    __shared__ typename P::T sorted_val[P::kStaged ? kSortTile : 1];

    for (uint32_t k = threadIdx.x; k < kSortWaves * kR; k += kSortThreads) (&wave_count[0][0])[k] = 0;
    // A tile waits for the tiles before it.  When the whole grid is certainly resident at the same time (use_ticket false:
    // the host has checked the grid against the kernel's real occupancy and against the other sorts in flight) nobody waits
    // for a workgroup that cannot start, whatever the dispatch order; otherwise tiles are handed out in start order through a
    // ticket counter (one serialising atomic per tile), so that every tile waited for has started.
    if (threadIdx.x == 0) { tile_fault = 0u; if (use_ticket) my_tile = atomicAdd(&tile_ticket[pass], 1u); }
    __syncthreads();
    const uint32_t tile = use_ticket ? my_tile : blockIdx.x;
    if (tile >= tiles) return false;   // (only a ticket can say so: grids that take their tile from blockIdx have exactly `tiles` workgroups)
    const uint32_t shift = pass * kBits;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint64_t lanes_below = (1ull << lane) - 1ull;

    uint64_t key[kSortItems];
    typename P::T val[kSortItems];
    uint32_t rank[kSortItems], from[kSortItems];
#pragma unroll
    for (uint32_t j = 0; j < kSortItems; j++) {
        const size_t i = tile_index(tile, wave, j, lane);
        const bool live = i < n;
        key[j] = live ? keys_in[i] : ~0ull;
        val[j] = live ? (pay_in ? pay_in[i] : P::first(first_src, i)) : P::none();
    }
#pragma unroll
    for (uint32_t j = 0; j < kSortItems; j++) {
        const bool live = tile_index(tile, wave, j, lane) < n;
        const uint32_t digit = uint32_t(key[j] >> shift) & (kR - 1);
        uint64_t same = __ballot(live);  // lanes of this wave holding the same digit (one ballot per digit bit)
#pragma unroll
        for (uint32_t b = 0; b < uint32_t(kBits); b++) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t vote = __ballot(bit);
            same &= bit ? vote : ~vote;
        }
        // the leader of every digit's lanes adds their number to the wave's count of that digit; what it gets back is asked for below,
        // all eight steps' answers together: waited for step by step -- returning LDS add, wait, ds_bpermute, wait -- they were sixteen
        // exposed LDS round trips per tile (a wave's LDS operations are performed in issue order: a later step's add finds an earlier one's)
        const uint32_t leader = live ? uint32_t(__ffsll((unsigned long long)same)) - 1u : lane;
        rank[j] = 0u;
        if (live && lane == leader) rank[j] = atomicAdd(&wave_count[wave][digit], uint32_t(__popcll(same)));
        from[j] = (leader << 8) | (live ? uint32_t(__popcll(same & lanes_below)) : 0u);   // whom to ask, and this key's place among the step's keys of its digit
    }
#pragma unroll
    for (uint32_t j = 0; j < kSortItems; j++)   // rank among this wave's keys of the same digit, in (j, lane) order
        rank[j] = uint32_t(__shfl(int(rank[j]), int(from[j] >> 8))) + (from[j] & 0xffu);
    __syncthreads();

    // thread t owns digits t + q * 256 from here on
    // this pass's words: one row of kR per tile (the tile's own digit counts), then one row per group of kGroupTiles tiles
    // (the group's digit counts)
    const uint32_t groups = radix_groups(tiles);
    uint32_t* region = tile_status + size_t(pass) * (tiles + groups) * kR;
    uint32_t count[kPer], local_excl[kPer], first_slot[kPer];
#pragma unroll
    for (uint32_t q = 0; q < kPer; q++) {
        const uint32_t d = threadIdx.x + q * kSortThreads;
        uint32_t c = 0;
#pragma unroll
        for (uint32_t w = 0; w < kSortWaves; w++) { const uint32_t v = wave_count[w][d]; wave_count[w][d] = c; c += v; }
        count[q] = c;
        __hip_atomic_store(&region[size_t(tile) * kR + d], kStatusPublished | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // exclusive scans over the digits (digit order: q-major, then thread): of the tile's counts (position of a digit's run in the
    // locally sorted tile) and of the whole input's counts (first output slot of the digit; the keys kernel / k_radix_bases left the
    // per-digit totals)
    uint32_t carry = 0, carry_all = 0;
#pragma unroll
    for (uint32_t q = 0; q < kPer; q++) {
        const uint32_t incl = wave_inclusive_sum(count[q]), incl_all = wave_inclusive_sum(total[q]);   // (on DPP: twelve ds_bpermute round trips in every tile's chain before)
        const uint32_t l = threadIdx.x & 63;
        if (q) __syncthreads();   // (the previous round's wave totals have been read)
        if (l == 63) { wave_total[threadIdx.x >> 6] = incl; wave_total_all[threadIdx.x >> 6] = incl_all; }
        __syncthreads();
        local_excl[q] = carry + incl - count[q]; first_slot[q] = carry_all + incl_all - total[q];
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) { local_excl[q] += wave_total[w]; first_slot[q] += wave_total_all[w]; }
        for (uint32_t w = 0; w < kSortWaves; w++) { carry += wave_total[w]; carry_all += wave_total_all[w]; }
    }

    // Keys of a digit in all earlier tiles, in two levels.  The tiles of a launch start together, so a chained look-back
    // finds only un-summed counts for a long way back (the nearest running total is ~4 r^2 tiles away after r round
    // trips: ~11 dependent round trips for 489 tiles, 14 us of every pass).  Instead: the last tile of every group of
    // kGroupTiles sums its group's rows and publishes the group's counts; a tile adds the counts of the lower groups and
    // of the lower tiles of its own group -- three dependent round trips whatever the number of tiles.  Every wait is for
    // a lower tile, as before.
    uint32_t polls = 0;
    auto sum_rows = [&](const uint32_t* first_row, uint32_t rows, uint32_t d) {   // sum of word d of `rows` consecutive rows, waiting for each to be published
        uint32_t sum = 0;
        for (uint32_t r0 = 0; r0 < rows; r0 += kLookback) {
            uint32_t word[kLookback];
#pragma unroll
            for (uint32_t k = 0; k < kLookback; k++)
                word[k] = r0 + k < rows ? __hip_atomic_load(first_row + size_t(r0 + k) * kR + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kStatusPublished;
#pragma unroll
            for (uint32_t k = 0; k < kLookback; k++) {
                uint32_t w = word[k];
                while ((w & kStatusPublished) == 0) {
                    if (++polls > kSpinLimit) { tile_fault = 1u; w = kStatusPublished; break; }   // give up: reported, never a hang
                    w = __hip_atomic_load(first_row + size_t(r0 + k) * kR + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                sum += w & kStatusValue;
            }
        }
        return sum;
    };
    const uint32_t group = tile / kGroupTiles, in_group = tile % kGroupTiles;
    uint32_t earlier[kPer];
#pragma unroll
    for (uint32_t q = 0; q < kPer; q++) {
        const uint32_t d = threadIdx.x + q * kSortThreads;
        earlier[q] = sum_rows(region + size_t(group) * kGroupTiles * kR, in_group, d);                 // lower tiles of this group
        if (in_group == kGroupTiles - 1 || tile == tiles - 1)                                           // last tile of the group: its counts, for the higher groups
            __hip_atomic_store(&region[size_t(tiles + group) * kR + d], kStatusPublished | (earlier[q] + count[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (uint32_t q = 0; q < kPer; q++) {
        const uint32_t d = threadIdx.x + q * kSortThreads;
        earlier[q] += sum_rows(region + size_t(tiles) * kR, group, d);                                  // lower groups
        offset[d] = int(first_slot[q] + earlier[q]) - int(local_excl[q]);
        // wave_count[w][d] becomes the local sorted position of wave w's first key of digit d
#pragma unroll
        for (uint32_t w = 0; w < kSortWaves; w++) wave_count[w][d] += local_excl[q];
    }
    __syncthreads();
    if (tile_fault) {   // a predecessor never published: the slots are unknown, write nothing (the sort's output is incomplete and says so)
        if (threadIdx.x == 0 && fault) atomicOr(fault, kSortFaultStalled);
        return true;
    }

#pragma unroll
    for (uint32_t j = 0; j < kSortItems; j++) {
        if (tile_index(tile, wave, j, lane) < n) {
            const uint32_t digit = uint32_t(key[j] >> shift) & (kR - 1);
            const uint32_t s = wave_count[wave][digit] + rank[j];
            sorted_key[s] = key[j];
            if (P::kStaged) sorted_val[s] = val[j];
            else pay_out[uint32_t(offset[digit] + int(s))] = val[j];
        }
    }
    __syncthreads();
    const uint32_t live_keys = min(uint32_t(kSortTile), n - min(n, tile * uint32_t(kSortTile)));
    for (uint32_t s = threadIdx.x; s < live_keys; s += kSortThreads) {
        const uint64_t k = sorted_key[s];
        const uint32_t slot = uint32_t(offset[uint32_t(k >> shift) & (kR - 1)] + int(s));
        keys_out[slot] = k;
        if (P::kStaged) pay_out[slot] = sorted_val[s];
    }
    return true;
}

// One launch = pass `first_pass`, and -- when `to_limit` is set -- every further pass up to *pass_limit as well, separated
// by grid-wide barriers.  The host enqueues as many launches as the previous frames needed passes (a hint read from
// host-mapped memory) and sets to_limit on the last of them: normally that launch does its one pass like the others, and
// when a frame suddenly needs more passes than were enqueued (its extent has grown) the rest still runs, just slower --
// instead of five surplus launches per frame that return at once.  to_limit requires every workgroup of the grid to be
// resident at once beside everything else in flight (tiles <= RadixSortBuffers::resident; radix_sort_enqueue_as checks).
template <class P, int kBits = kRadixBits>
__global__ void __launch_bounds__(kSortThreads, 4)   // (<= 128 registers: four waves per SIMD, as before the ranking's answers were asked for together)
k_radix_onesweep(uint64_t* __restrict__ keys0, typename P::T* __restrict__ pay0, uint64_t* __restrict__ keys1, typename P::T* __restrict__ pay1,
                 uint32_t n, uint32_t first_pass, uint32_t to_limit, uint32_t tiles, const uint32_t* __restrict__ digit_base, uint32_t base_stripes,
                 uint32_t* __restrict__ tile_status, uint32_t* __restrict__ tile_ticket, const uint32_t* __restrict__ pass_limit, const void* __restrict__ first_src,
                 uint32_t use_ticket, RadixSortSignals sig) {
    const uint32_t limit = pass_limit ? min(uint32_t(kMaxPasses), *pass_limit) : uint32_t(kMaxPasses);
    uint32_t* arrived = tile_ticket + kMaxPasses;   // zeroed with the look-back table
    (void)kBits;
    if (sig.dep) {   // (RadixSortSignals::dep; such a sort has a launch per pass)
        if ((first_pass == kDepFirstPass || first_pass == kDepFirstPass + 1u) && sig.dep[2 * kRadix] == 0u) return;
        if (first_pass == kDepTopPass && blockIdx.x == 0) for (uint32_t k = threadIdx.x; k < kDepWords; k += kSortThreads) sig.dep[k] = 0u;   // for the next sort
    }
    // this launch runs behind pass 0 (stream order): the caller's raw points have been read for the last time
    if (sig.consumed && first_pass > 0u && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(sig.consumed, sig.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (uint32_t pass = first_pass; pass < limit; pass++) {
        const bool odd = pass & 1u;
        radix_onesweep_tile<P, kBits>(odd ? keys1 : keys0, pass == 0 ? nullptr : (odd ? pay1 : pay0), odd ? keys0 : keys1, odd ? pay0 : pay1, n, pass, tiles, digit_base, base_stripes,
                               tile_status, tile_ticket, first_src, use_ticket != 0u, sig.fault);
        if (!to_limit || pass + 1 >= limit) break;
        // grid barrier: this pass's output (plain stores) must be visible to every workgroup's loads of the next pass
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(arrived, 1u);
            const uint32_t target = gridDim.x * (pass - first_pass + 1u);
            uint32_t polls = 0;
            while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++polls > kSpinLimit) { if (sig.fault) atomicOr(sig.fault, kSortFaultStalled); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            __threadfence();
        }
        __syncthreads();
    }
}

// The catch-all behind the hinted launches of a sort whose grid is too large to be resident at once (a 1 M-point frame with three
// frames in flight: 489 tiles against a share of 320 workgroups): passes first_pass .. limit - 1 -- normally none -- by as many
// workgroups as ARE resident at once (`participants`: the host's share), each drawing tiles by ticket until the pass has none, with a
// barrier among them behind every pass, the last one included.  Until round 4 a launch of its own that returned at once in
// nearly every frame -- 5 us in the chain of every large frame, 1.3 % of the frame rate (CHAD_DEBUG_NO_REST experiment); now the
// prologue of the kernel that follows the sort (k_frame_gather): RadixRest carries what it needs, participants == 0 says "no
// catch-all: the grid was resident, or every pass got its launch".
struct RadixRest {
    uint64_t* keys[2]; uint4* points[2];
    uint32_t n, first_pass, tiles, base_stripes, participants;
    const uint32_t* digit_base;
    uint32_t* tile_status; uint32_t* tile_ticket;
    const void* first_src;
    uint32_t* fault;
};
// words behind the per-pass tickets (all zeroed with the look-back table before every sort)
constexpr uint32_t kRestArrived = kMaxPasses, kRestGate = kMaxPasses + 1, kRestDone = kMaxPasses + 2;

// Every workgroup of the calling grid enters with the same `limit` > r.first_pass.  The first r.participants to ARRIVE do the passes (a
// workgroup that has arrived is running: nobody waits for one that cannot start); the others wait for the word that says so.
template <int kBits>
__device__ __forceinline__ void radix_rest_points(const RadixRest& r, uint32_t limit) {
    __shared__ uint32_t role_;
    if (threadIdx.x == 0) role_ = atomicAdd(&r.tile_ticket[kRestGate], 1u);
    __syncthreads();
    const uint32_t role = role_;
    uint32_t* done = &r.tile_ticket[kRestDone];
    if (role >= r.participants) {
        if (threadIdx.x == 0) {
            uint32_t polls = 0;
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                if (++polls > kSpinLimit) { if (r.fault) atomicOr(r.fault, kSortFaultStalled); break; }
                __builtin_amdgcn_s_sleep(16);
            }
            __threadfence();
        }
        __syncthreads();
        return;
    }
    uint32_t* arrived = &r.tile_ticket[kRestArrived];
    for (uint32_t pass = r.first_pass; pass < limit; pass++) {
        const bool odd = pass & 1u;
        while (radix_onesweep_tile<PointPayload, kBits>(r.keys[odd], pass == 0 ? nullptr : r.points[odd], r.keys[odd ^ 1u], r.points[odd ^ 1u], r.n, pass, r.tiles, r.digit_base,
                                                        r.base_stripes, r.tile_status, r.tile_ticket, r.first_src, true, r.fault))
            __syncthreads();   // the tile's LDS is read to the end before the next one overwrites it
        // barrier among the participants: this pass's output (plain stores) must be visible to every workgroup's loads of the next
        // pass -- and, behind the last pass, to the caller
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(arrived, 1u);
            const uint32_t target = r.participants * (pass - r.first_pass + 1u);
            uint32_t polls = 0;
            while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++polls > kSpinLimit) { if (r.fault) atomicOr(r.fault, kSortFaultStalled); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            __threadfence();
        }
        __syncthreads();
    }
    if (role == 0u && threadIdx.x == 0) __hip_atomic_store(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// launch_passes (0 = all `passes`): launches to enqueue; the last of them covers whatever passes remain (k_radix_onesweep,
// to_limit) when the grid is small enough to be resident at once -- otherwise every pass gets its launch.
// P = IndexPayload: pairs (key, input position) in b.keys / b.idx.  P = PointPayload: (key, point) in b.keys / b.points,
// the first pass reading the raw points `first_src` (float xyz[n][3]) in input order.
template <class P>
inline void radix_sort_enqueue_as(const RadixSortBuffers& b, typename P::T* const pay[2], uint32_t n, uint32_t passes, const uint32_t* pass_limit, hipStream_t stream,
                                  uint32_t hist_blocks, uint32_t launch_passes, const void* first_src,
                                  const RadixSortSignals& sig = RadixSortSignals{}, uint32_t digit_bits = kRadixBits, RadixRest* rest_out = nullptr) {
    if (rest_out) *rest_out = RadixRest{};
    if (n == 0) return;
    // RadixSortSignals::dep (passes 5 and 6 skipped when their digits are functions of the top digit) is only sound for a sort that takes
    // its own histograms (they compute the dep words), runs all eight passes and gives every pass its own launch (pass 7's launch
    // re-zeroes the words): any other caller's request is dropped here rather than trusted
    RadixSortSignals checked = sig;
    if (checked.dep && !(hist_blocks == 0 && passes == uint32_t(kMaxPasses) && !(pass_limit && launch_passes && launch_passes < passes))) checked.dep = nullptr;
    const RadixSortSignals& sig_ = checked;
    const uint32_t tiles = radix_tiles(n);
    const bool all_resident = b.resident != 0u && tiles <= b.resident;
    uint32_t* ticket = b.tile_status + radix_status_words(tiles) - 64;
    if (hist_blocks == 0) {
        hipLaunchKernelGGL(k_radix_histograms, dim3(kHistBlocks), dim3(kSortThreads), 0, stream, b.keys[0], n, passes, pass_limit, b.hist_part, b.tile_status,
                           uint32_t(radix_status_words(tiles)), sig_.dep);
    }
    // hist_blocks != 0: the caller's keys kernel has already added its digit counts into the striped totals (k_frame_keys)
    const uint32_t base_stripes = hist_blocks ? kBaseStripes : 1u;
    if (!hist_blocks) hipLaunchKernelGGL(k_radix_bases, dim3(passes * 4), dim3(4 * kRadix), 0, stream, b.hist_part, uint32_t(kHistBlocks), passes, pass_limit, b.digit_base);
    const bool hinted = pass_limit && launch_passes && launch_passes < passes && all_resident;
    // a grid too large to be resident at once: the hinted launches; whatever the frame needs beyond them is left to the caller's next kernel (RadixRest)
    // (the catch-all's barrier needs its participants resident at once INSIDE the caller's next kernel: that kernel's own share of the
    // chip bounds them too -- RadixSortBuffers::rest_limit, 0 = nobody can host a catch-all: every pass gets its launch)
    const uint32_t rest_participants = std::min(std::min(tiles, b.resident), b.rest_limit);
    const bool with_rest = pass_limit && launch_passes && launch_passes < passes && !all_resident && rest_participants != 0u && rest_out != nullptr;   // (nobody to take the rest: every pass gets its launch)
    const uint32_t launches = hinted || with_rest ? launch_passes : passes;
    for (uint32_t p = 0; p < launches; p++) {
        const bool to_limit = hinted && p + 1 == launches;
        // A launch that may end up behind grid barriers shares the chip with the like launches of the other frames in flight: each
        // of them is at most RadixSortBuffers::resident workgroups -- the host's SHARE of what the device holds at once -- so all of
        // them fit together, and whatever else occupies the CUs meanwhile finishes without waiting for a sort.  (Rounds 1-2 made
        // these launches take turns through an event: with four or more frames in flight every frame's last pass then queued
        // behind the previous frame's -- 36-57 us of waiting per 200 k-point frame, scripts/timeline.py -- for a barrier that a
        // frame reaches only when its key has outgrown the hint.)
        auto launch = [&](auto kernel) {
            if (sig_.pass_events)
                hipExtLaunchKernelGGL(kernel, dim3(tiles), dim3(kSortThreads), 0, stream, sig_.pass_events[2 * p], sig_.pass_events[2 * p + 1], 0u, b.keys[0], pay[0], b.keys[1], pay[1], n, p,
                                      to_limit ? 1u : 0u, tiles, b.digit_base, base_stripes, b.tile_status, ticket, pass_limit, first_src, all_resident ? 0u : 1u, sig_);
            else
                hipLaunchKernelGGL(kernel, dim3(tiles), dim3(kSortThreads), 0, stream, b.keys[0], pay[0], b.keys[1], pay[1], n, p,
                                   to_limit ? 1u : 0u, tiles, b.digit_base, base_stripes, b.tile_status, ticket, pass_limit, first_src, all_resident ? 0u : 1u, sig_);
        };
        if (digit_bits == uint32_t(kWideBits)) launch(k_radix_onesweep<P, kWideBits>); else launch(k_radix_onesweep<P, kRadixBits>);
    }
    if (with_rest) {   // the caller's next kernel finishes the sort if it has to (radix_rest_points): no launch here
        if constexpr (std::is_same<P, PointPayload>::value) {
            if (rest_out) {
                rest_out->keys[0] = b.keys[0]; rest_out->keys[1] = b.keys[1]; rest_out->points[0] = pay[0]; rest_out->points[1] = pay[1];
                rest_out->n = n; rest_out->first_pass = launches; rest_out->tiles = tiles; rest_out->base_stripes = base_stripes;
                rest_out->participants = rest_participants;
                rest_out->digit_base = b.digit_base; rest_out->tile_status = b.tile_status; rest_out->tile_ticket = ticket; rest_out->first_src = first_src; rest_out->fault = sig_.fault;
            }
        }
    }
    if (sig_.launches_out) *sig_.launches_out = launches;
}
inline void radix_sort_enqueue(const RadixSortBuffers& b, uint32_t n, uint32_t passes, const uint32_t* pass_limit, hipStream_t stream, uint32_t hist_blocks = 0,
                               uint32_t launch_passes = 0, const RadixSortSignals& sig = RadixSortSignals{}) {
    radix_sort_enqueue_as<IndexPayload>(b, b.idx, n, passes, pass_limit, stream, hist_blocks, launch_passes, nullptr, sig);
}
inline void radix_sort_points_enqueue(const RadixSortBuffers& b, const float* xyz, uint32_t n, uint32_t passes, const uint32_t* pass_limit, hipStream_t stream,
                                      uint32_t hist_blocks, uint32_t launch_passes, const RadixSortSignals& sig, uint32_t digit_bits, RadixRest* rest_out) {
    radix_sort_enqueue_as<PointPayload>(b, b.points, n, passes, pass_limit, stream, hist_blocks, launch_passes, xyz, sig, digit_bits, rest_out);
}
// Workgroups of a pass kernel the device holds at once: the kernel's real occupancy (registers, LDS) x CU count, with a quarter
// held back (the runtime's answer can exceed what the hardware admits by one per CU -- MI355X_MICROARCH.md, Residency).
template <class P>
inline uint32_t radix_resident_workgroups() {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    int per_cu_wide = 0;   // (the wider digit holds more LDS: the smaller of the two answers)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_radix_onesweep<P, kRadixBits>, kSortThreads, 0) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_wide, k_radix_onesweep<P, kWideBits>, kSortThreads, 0) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
        hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 0u; }
    per_cu = per_cu < per_cu_wide ? per_cu : per_cu_wide;
    return uint32_t(per_cu > 0 ? per_cu : 0) * uint32_t(prop.multiProcessorCount) * 3u / 4u;
}

}  // namespace chadhip

// map_kernels.hpp -- the sparse voxel-block map in HBM and the TSDF update kernels.
//
// Replaces Octree::insert(MortonCode) + _node_lookup (octree.hpp:31-78,187) and the weighted update of
// octree.hpp:153-164.
//
// Layout in HBM (structure of arrays, one "brick" per 4x4x4-voxel block, voxel index = low 6 code bits).  A brick
// lives AT its hash slot -- there is no separate brick pool -- so that finding or creating a block is ONE
// compare-and-swap on its key (one memory round trip; the first version needed four dependent ones: key load,
// CAS, pool counter, index publish/wait -- 17 us of the update kernel):
//   hash_keys[H]     u64  open-addressed table of block keys (code & ~63); EMPTY = ~0; linear probing, so a probe
//                         sequence walks consecutive 8-byte keys of one 128-byte line before touching the next
//   brick_sum[H*64]  i64  sum of the clamped signed distances observed by each voxel of the brick in slot s, in units of
//                         trunc / 2^30 (exact, order-independent; see kSumOne below)
//   brick_cnt[H*64]  u32  number of observations (the reference's Leaf::_weight)
//   brick_pack[H*64] u64  staging word of the scatter form: count << 40 + sum in units of trunc / 2^16, added with ONE atomic per
//                         observation; every reader adds it to (brick_sum, brick_cnt) on the fly (voxel_totals), k_fold_packed moves
//                         it over before the count field could run out (see kPack... below)
//   live_bits[H/32]  u32  bit s set: slot s holds a brick (set fire-and-forget by the claiming lane); clearing,
//                         export and growth walk this bitmap -- O(H/32 + bricks), not O(table)
//   brick_count[kCountStripes*16] u32  number of bricks, striped over kCountStripes = 256 cache lines (no same-address atomics)
// H = 2 * capacity rounded up to a power of two, i.e. the table -- and with it the brick storage, 768 B per slot --
// is at most half full at the nominal capacity; 288 GB of HBM make that trade (2 x 1288 B per block) a cheap one.
// Slots that hold no brick keep all-zero rows (zeroed at allocation and again whenever a brick is released).
// The reference keeps a running mean sd and a count w per leaf, with unit weight per observation
// (octree.hpp:161-163); (sum, cnt) is the same information with sd = sum / cnt, and it is what makes the update a
// commutative reduction: LDS partial sums, atomics across workgroups and an all-reduce across GPUs all combine
// by addition.
#pragma once
#include "device_math.hpp"
#include "frame_kernels.hpp"

namespace chadhip {

constexpr uint64_t kEmptyKey = ~0ull;
// MapView::overflow, one word per table: bit 0 = a block found no slot (sticky until a reset); bit 2 = the table's exact rows
// (brick_sum, brick_cnt) have been written since it was last cleared.  A table that only ever took the scatter form's staging
// words (sparse frames: BASELINE config 5) has all-zero exact rows, and everything that walks the live bricks -- the archive's
// compaction, the exchange's, the clear -- then touches 512 of a slot's 1 288 bytes (1.65 M live bricks per cfg5 submap: 2.1 GB
// less to read and 1.3 GB less to zero per roll-over).  Raised by one lane per launch (or per rare event), never per observation.
constexpr uint32_t kTableOverflowed = 1u, kTableExactRows = 4u;
constexpr uint32_t kNoBrick = 0xfffffffeu;   // table too full: updates to this block are dropped (overflow flag set)
constexpr uint32_t kMaxProbes = 256;         // at load <= 0.5 probe sequences are a handful of slots long

// A voxel of the ACTIVE map holds (sum of its observations' clamped distances, their count) -- the same information as the
// reference's running mean and weight (octree.hpp:161-163).  The sum is a 64-bit integer in units of trunc / 2^30 (the
// fixed point the update kernel's LDS tile accumulates in): integer addition is exact and does not depend on the order of
// the atomics, so a map is a function of its frames alone -- bit-reproducible -- and a voxel observed a hundred thousand
// times still holds the exact mean (f32 sums of that many terms, like the reference's f32 running mean, drift by 1e-4 of
// the truncation distance and more).  Archives, exports and the multi-GPU exchange carry f32 sums: k_compact_bricks
// converts, with one rounding.  The reference's NaN leaves (a degenerate neighbourhood's NaN normal, Q4) are kept as a
// flag in the count's top bit: the count goes on counting, the compacted sum comes out NaN.
constexpr float kSumOne = 1073741824.0f;        // 2^30 units per truncation distance
constexpr uint32_t kPoisonedCount = 0x80000000u;
__device__ __forceinline__ long long sum_units(float d, float units_per_metre) { return (long long)rintf(d * units_per_metre); }

// The scatter form of the update (sparse frames: about one observation per voxel and frame) is bound by the number of memory-side
// atomic requests -- on this chip every device-scope atomic travels to the memory side, 64 bytes per request whatever it carries:
// 222 us per cfg5 frame with two atomics per observation (sum, count), 147 us with one, 62 us with none (scripts/cfg5_costs.py).
// So an observation of the scatter form is ONE 64-bit add into a staging word per voxel: (1 << 40) + round(d * 2^16 / trunc) --
// count in the top 24 bits, the signed sum below it (a negative sum borrows from the count; decoding undoes that).  2^16 units per
// truncation distance: 7.6e-6 of it per observation at worst, against the 1e-4 bar, and still an exact, order-independent integer.
// The sum field holds 2^23 observations of +-trunc (kPackCapacity): the host folds the staging words into (brick_sum, brick_cnt)
// before a voxel could have received that many (chad_hip.hip, fold_if_due), and only frames of at most kPackMaxPoints points use
// them.  Between the executions of two folds a voxel gets at most one observation per point of: the frames counted towards the
// second fold, the frame that made it due, the frames that overtake it on other streams (fewer than the frame slots), and the
// frames enqueued before the first fold that ran behind it (two per slot at most) -- so a fold is due after
// pack_fold_points(slots) points.
constexpr uint32_t kPackCountShift = 40;
constexpr float kPackOne = 65536.0f;                         // 2^16 units per truncation distance
constexpr long long kPackToSum = 1ll << (30 - 16);           // one staging unit in units of the exact sum (kSumOne = 2^30)
constexpr uint32_t kPackMaxPoints = 1u << 18;                // larger frames take two exact atomics per observation in the scatter form
constexpr uint64_t kPackCapacity = 1ull << 23;               // observations of +-trunc the staging word's sum field holds
inline uint64_t pack_fold_points(uint32_t slots) { return kPackCapacity - uint64_t(3u * slots + 2u) * kPackMaxPoints; }   // 5.5 M at three slots, 1.6 M at eight

struct MapView {
    uint64_t* hash_keys;
    long long* brick_sum;
    uint32_t* brick_cnt;
    unsigned long long* brick_pack;
    uint32_t* live_bits;
    uint32_t* brick_count;   // striped device counter of live bricks
    uint32_t* overflow;      // sticky: set when a block found no slot (its updates are dropped)
    uint32_t hash_mask;      // H - 1 (H a power of two)
    uint32_t brick_capacity; // nominal capacity (H / 2 or less): the pools grow once half of it is in use
};

__device__ __forceinline__ bool table_pack_only(const MapView& m) { return (__hip_atomic_load(m.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kTableExactRows) == 0u; }
__device__ __forceinline__ void mark_exact_rows(const MapView& m) { atomicOr(m.overflow, kTableExactRows); }

// One observation into cell `at` of the active map (the scatter form; the tiled update adds whole partial sums).
__device__ __forceinline__ void add_observation(const MapView& m, size_t at, float d, float units_per_metre) {
    if (d == d) atomicAdd(reinterpret_cast<unsigned long long*>(&m.brick_sum[at]), (unsigned long long)sum_units(d, units_per_metre));
    else atomicOr(&m.brick_cnt[at], kPoisonedCount);   // NaN (Q4): the voxel's mean is NaN from here on, as in the reference
    atomicAdd(&m.brick_cnt[at], 1u);
}

// The scatter form's observation: one atomic (see kPackCountShift).  A NaN distance (Q4) counts and poisons as above.
__device__ __forceinline__ void add_observation_packed(const MapView& m, size_t at, float d, float pack_units_per_metre) {
    unsigned long long word = 1ull << kPackCountShift;
    if (d == d) word += (unsigned long long)(long long)rintf(d * pack_units_per_metre);
    else { atomicOr(&m.brick_cnt[at], kPoisonedCount); mark_exact_rows(m); }
    atomicAdd(&m.brick_pack[at], word);
}
__device__ __forceinline__ void unpack_staged(unsigned long long word, long long* sum_units_out, uint32_t* count_out) {
    const long long staged = (long long)(word << (64 - kPackCountShift)) >> (64 - kPackCountShift);   // the low 40 bits, sign-extended
    *count_out = uint32_t((word - (unsigned long long)staged) >> kPackCountShift);
    *sum_units_out = staged * kPackToSum;
}
// What voxel `at` holds: the exact sums plus whatever is still staged.  count keeps the poison flag in its top bit.
// pack_only (wave-uniform, table_pack_only() read once per kernel behind the table's last update): the exact rows are known to be zero
__device__ __forceinline__ void voxel_totals(const MapView& m, size_t at, long long* sum, uint32_t* count, bool pack_only = false) {
    long long s = pack_only ? 0ll : m.brick_sum[at];
    uint32_t c = pack_only ? 0u : m.brick_cnt[at];
    const unsigned long long word = m.brick_pack[at];
    if (word) {
        long long staged_sum; uint32_t staged_count;
        unpack_staged(word, &staged_sum, &staged_count);
        s += staged_sum; c += staged_count;   // (the count never reaches the flag bit: 2^31 observations)
    }
    *sum = s; *count = c;
}

// Home slot of a block: a multiplicative hash of its Morton index -- on purpose WITHOUT any locality.  Keeping the block's place in
// its 2x2x2 group (eight neighbouring blocks in eight consecutive slots: their keys in one line, their rows in one 4 KB stretch)
// looked attractive for the sparse update's 2.6 M random accesses per frame and was measured (round 4, scripts/ab_headline.py on one
// box): the sparse update alone 97 -> 118 us, cfg5 0.138 -> 0.143 ms per frame, cfg3 erratic (0.051 -> 0.051-0.065), cfg2 1 % faster:
// neighbouring rays' atomics then queue at the same memory channel instead of spreading over all of them, and linear probing
// clusters (simulated on a cfg5 submap, 1.65 M blocks in 2^23 slots: mean / 99th percentile / longest probe sequence 1.10 / 3 / 13
// for this hash, 1.39 / 11 / 51 for groups of 8, 3.5 / 60 / 216 for groups of 64, 2.2 / 32 / 510 with the whole index kept).
__device__ __forceinline__ uint32_t hash_block(uint64_t key, uint32_t mask) {
    return uint32_t(((key >> 6) * 0x9E3779B97F4A7C15ull) >> 32) & mask;
}

__device__ __forceinline__ uint32_t live_brick_count(const MapView& m) {
    uint32_t n = 0;
    for (uint32_t k = 0; k < kCountStripes; k++) n += m.brick_count[k * 16];
    return n;
}

// Slot (= brick index) of `key`, claiming an empty slot if the block is new.  One CAS per probe; a claim also sets
// the slot's live bit and bumps a counter stripe, both without waiting for a result.  Returns kNoBrick when no
// slot is found within kMaxProbes.
__device__ __forceinline__ uint32_t find_or_create_brick(const MapView& m, uint64_t key, uint32_t slot) {
    for (uint32_t probes = 0; probes < kMaxProbes; probes++) {
        const uint64_t prev = atomicCAS((unsigned long long*)&m.hash_keys[slot], (unsigned long long)kEmptyKey, (unsigned long long)key);
        if (prev == key) return slot;
        if (prev == kEmptyKey) {
            atomicOr(&m.live_bits[slot >> 5], 1u << (slot & 31u));
            atomicAdd(&m.brick_count[(slot & (kCountStripes - 1)) * 16], 1u);
            return slot;
        }
        slot = (slot + 1) & m.hash_mask;
    }
    atomicOr(m.overflow, 1u);
    return kNoBrick;
}
__device__ __forceinline__ uint32_t find_or_create_brick(const MapView& m, uint64_t key) {
    return find_or_create_brick(m, key, hash_block(key, m.hash_mask));
}

// ---- update, general form: one lane per point, global atomics per observation (octree.hpp:86-164) -----------
// The body is shared: k_update_scatter runs it alone; k_update_tiled (tile_update.hpp) runs it instead of its own loop when
// the frame turns out to be sparse, so that the automatic mode needs one launch, not two.
// `pack_counter` / `pack_seen_host`: a device counter of the launches that have used the staging words, and its mirror in
// host-mapped memory, by which the host learns that there is something to fold (chad_hip.hip, fold_if_due).
//
// What bounds this form is the number of memory-side atomic REQUESTS, and a request is a 64-byte segment touched by one
// instruction of one wave -- the eight staging words of a 2x2x2 octant of a brick.  With a lane per ray, the lanes of an
// instruction are sixty-four different rays at the same step: sixty-four requests.  So the walk only NOTES its observations, in
// LDS (`stage`: this wave's rows of kStageRow entries, one row per step), and a second phase issues them transposed: sixteen
// lanes are the steps of ONE ray, four neighbouring rays per instruction -- the two or three voxels a ray visits in an octant,
// and what the neighbouring rays of a sorted frame add to the same octants, travel as one request.  Steps beyond the staged rows
// (stage_rows, whatever LDS the caller has) and frames too large for the staging words go straight to memory as before.
// (Round 4 also built this form with the walk, ALL block look-ups of a batch and then the atomics in separate phases -- three or four
// dependent round trips per wave instead of a dozen: 97 us alone either way, the same frame rate in every cfg5 mode on one box.  The
// kernel is bound by the memory side's rate of atomic requests, 1.6 M at ~20 G/s, not by a wave's chain of look-ups.)
constexpr uint32_t kStageRow = 65;   // entries of a row: one per lane, and one of padding (the transposed reads stride over rows)
__device__ __forceinline__ void scatter_update(const float4* __restrict__ pts, const float4* __restrict__ normals, FrameHeader* __restrict__ h,
                                               F3 pos, float res, float recip, float trunc, const MapView& m, uint32_t* pack_counter, uint32_t* pack_seen_host,
                                               unsigned long long* stage, uint32_t stage_rows) {
    const uint32_t n = h->n_usable;
    const bool packed = n <= kPackMaxPoints;
    if (packed && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(pack_seen_host, atomicAdd(pack_counter, 1u) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!packed && blockIdx.x == 0 && threadIdx.x == 0) mark_exact_rows(m);   // two exact atomics per observation: the exact rows are in use
    const float pack_units = kPackOne / trunc, sum_units_per_metre = kSumOne / trunc;
    const uint32_t lane = threadIdx.x & 63u, rows = packed ? min(stage_rows, 16u) : 0u;
    constexpr unsigned long long kNothing = ~0ull;
    unsigned long long applied = 0;
    for (size_t base = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) & ~size_t(63); base < n; base += size_t(gridDim.x) * blockDim.x) {
        const size_t i = base + lane;
        uint32_t k = 0;   // steps of this lane's ray so far
        if (i < n) {
            const float4 q = pts[i], nq = normals[i];
            const F3 p{q.x, q.y, q.z}, nrm{nq.x, nq.y, nq.z};
            RayWalk w;
            w.begin(p, pos, res, recip, trunc);
            uint64_t cached_key = kEmptyKey;
            uint32_t brick = kNoBrick;
            do {
                const uint64_t mc = morton_encode(w.cur.x, w.cur.y, w.cur.z);
                const uint64_t key = mc & kBlockMask;
                if (key != cached_key) {
                    // most blocks of a frame are there already (the sensor moves a fraction of its range per frame): a plain load finds
                    // them, and only a miss pays for the compare-and-swap (a memory-side atomic like the adds below)
                    const uint32_t home = hash_block(key, m.hash_mask);
                    brick = __hip_atomic_load(&m.hash_keys[home], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key ? home : find_or_create_brick(m, key, home);
                    cached_key = key;
                }
                unsigned long long note = kNothing;
                if (brick != kNoBrick) {
                    const size_t at = size_t(brick) * 64 + (uint32_t(mc) & 63u);
                    const float d = observe(w.cur, p, nrm, res, trunc);
                    if (k < rows) {   // noted: [cell : 40 | distance in staging units : 24, two's complement]
                        if (!(d == d)) { atomicOr(&m.brick_cnt[at], kPoisonedCount); mark_exact_rows(m); }   // NaN (Q4): counted, adds nothing, poisons (the flag lives in the exact count row)
                        note = (uint64_t(at) << 24) | (uint32_t(d == d ? int(rintf(d * pack_units)) : 0) & 0xffffffu);
                    } else if (packed) add_observation_packed(m, at, d, pack_units);
                    else add_observation(m, at, d, sum_units_per_metre);
                    applied++;
                }
                if (k < rows) stage[k * kStageRow + lane] = note;
                k++;
            } while (w.next());
        }
        if (rows) {
            for (; k < rows; k++) stage[k * kStageRow + lane] = kNothing;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a wave's LDS operations complete in order: its own notes are there)
            const uint32_t step = lane & 15u, ray = lane >> 4;
            if (step < rows)
                for (uint32_t j = 0; j < 16u; j++) {
                    const unsigned long long note = stage[step * kStageRow + 4u * j + ray];
                    if (note != kNothing)
                        atomicAdd(&m.brick_pack[note >> 24], (1ull << kPackCountShift) + (unsigned long long)(long long)(int(uint32_t(note) << 8) >> 8));
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read before the next batch's notes overwrite them
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) applied += __shfl_xor(applied, o);
    if ((threadIdx.x & 63) == 0 && applied) atomicAdd(&h->n_updates[(blockIdx.x % kUpdateStripes) * 16u], applied);
}

__global__ void __launch_bounds__(256)
k_update_scatter(const float4* __restrict__ points0, const float4* __restrict__ points1, const float4* __restrict__ normals, FrameHeader* __restrict__ h,
                 F3 pos, float res, float recip, float trunc, MapView m, uint32_t* pack_counter, uint32_t* pack_seen_host) {
    const float4* __restrict__ pts = (h->sort_passes & 1u) ? points0 : points1;   // the frame in processing order: k_frame_normals' output (the sort's other buffer)
    __shared__ unsigned long long stage[4][16 * kStageRow];
    scatter_update(pts, normals, h, pos, res, recip, trunc, m, pack_counter, pack_seen_host, stage[threadIdx.x >> 6], 16u);
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

// Self-test of device_math.hpp's reciprocal against the IEEE quotient, over ALL 2^32 f32 bit patterns: out[0] inputs inside the range
// rcp_newton is used for, out[1] of those where it differs from 1.0f / x (must be 0), out[2] inputs -- any -- where recip1 differs
// (must be 0; NaN equals NaN), out[3] inputs outside the range where rcp_newton WOULD differ (why the range test is there)
__global__ void __launch_bounds__(256)
k_selftest_reciprocal(unsigned long long* __restrict__ out) {
    unsigned long long in_range = 0, bad_fast = 0, bad_any = 0, bad_outside = 0;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < (1ull << 32); i += uint64_t(gridDim.x) * blockDim.x) {
        const float x = __uint_as_float(uint32_t(i));
        const float want = 1.0f / x, fast = rcp_newton(x), got = recip1(x);
        const bool same_fast = __float_as_uint(fast) == __float_as_uint(want) || (fast != fast && want != want);
        if (rcp_newton_exact_for(x)) { in_range++; bad_fast += same_fast ? 0u : 1u; } else bad_outside += same_fast ? 0u : 1u;
        bad_any += (__float_as_uint(got) == __float_as_uint(want) || (got != got && want != want)) ? 0u : 1u;
    }
    const unsigned long long v[4] = {in_range, bad_fast, bad_any, bad_outside};
    for (int k = 0; k < 4; k++) {
        unsigned long long t = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if ((threadIdx.x & 63) == 0 && t) atomicAdd(&out[k], t);
    }
}

// Self-test of the wave-wide scans and reductions that run on the DPP network (radix_sort.hpp: wave_inclusive_sum; frame_kernels.hpp:
// wave_suffix_min, wave_min / wave_max / wave_sum / wave_min_u64) against plain loops over the same 64 values, `rounds` times per wave
// with values drawn from a hash of (wave, round, lane) -- some rounds all equal, some with the extreme in a single lane.
// out[k] = lanes whose result differed, k = 0 inclusive sum, 1 suffix minimum, 2 minimum, 3 maximum, 4 sum, 5 64-bit minimum; out[6] = lanes checked.
__global__ void __launch_bounds__(256)
k_selftest_wave_ops(unsigned long long* __restrict__ out, uint32_t rounds) {
    __shared__ uint32_t value[4][64];
    __shared__ unsigned long long wide[4][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, wave_global = blockIdx.x * 4u + wave;
    uint32_t bad[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t r = 0; r < rounds; r++) {
        uint64_t hsh = (uint64_t(wave_global) << 40) ^ (uint64_t(r) << 8) ^ lane;
        hsh ^= hsh >> 33; hsh *= 0xff51afd7ed558ccdull; hsh ^= hsh >> 33; hsh *= 0xc4ceb9fe1a85ec53ull; hsh ^= hsh >> 33;
        uint32_t v = uint32_t(hsh) >> ((r % 5u) * 6u);               // small and large magnitudes (the sum wraps: so does the loop's)
        if (r % 7u == 3u) v = 12345u;                                // all lanes equal
        if (r % 7u == 5u) v = lane == (r % 64u) ? 1u : 0xfffffff0u;  // the minimum in one lane only
        const unsigned long long w = (hsh << 11) ^ v;
        value[wave][lane] = v; wide[wave][lane] = w;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (a wave's own LDS writes, then its reads: in order)
        uint32_t incl = 0, suffix = 0xffffffffu, lo = 0xffffffffu, total = 0;
        int smin = 0x7fffffff, smax = int(0x80000000);
        unsigned long long lo64 = ~0ull;
        for (uint32_t l = 0; l < 64u; l++) {
            const uint32_t x = value[wave][l];
            if (l <= lane) incl += x;
            if (l >= lane) suffix = min(suffix, x);
            lo = min(lo, x); total += x;
            smin = min(smin, int(x)); smax = max(smax, int(x));
            lo64 = wide[wave][l] < lo64 ? wide[wave][l] : lo64;
        }
        (void)lo;
        bad[0] += wave_inclusive_sum(v) != incl;
        bad[1] += wave_suffix_min(v) != suffix;
        bad[2] += wave_min(int(v)) != smin;
        bad[3] += wave_max(int(v)) != smax;
        bad[4] += wave_sum(v) != total;
        bad[5] += wave_min_u64(w) != lo64;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int k = 0; k < 6; k++) if (bad[k]) atomicAdd(&out[k], (unsigned long long)bad[k]);
    atomicAdd(&out[6], (unsigned long long)rounds);
}

// Morton codes of a frame's points in processing order (float4 x, y, z, index): the stage read-back of chad_map_last_frame
__global__ void k_morton_codes_of_points(const float4* __restrict__ pts, uint32_t n, float recip, uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = pts[i];
    const I3 v = voxel_of(F3{q.x, q.y, q.z}, recip);
    out[i] = morton_encode(v.x, v.y, v.z);
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

// Device half of Submap::finalize (submap.hpp:70-99): the bricks of a closed submap, taken in the order given by
// `order` (ascending block key), become eight leaf clusters each -- octet o = cells 8o..8o+7 of the brick, one byte per
// leaf in Morton order.  Signed-distance byte: uint8(clamp(sd / trunc, -1, 1) * 127 + 127) (cluster.hpp:13-26), 0xff
// for a voxel never observed (cluster.hpp:28-31), so an octet without leaves comes out as all ones and is skipped by the
// host.  Weight byte: min(count, 254), again 0xff for none; only written when out_weight is given (the reference
// itself stores 0xff for every weight -- Q6, submap.hpp:93 -- which needs no array).
// One wave per brick: lane = cell, 2 x 256 B read, 64 B (+ 64 B) written.  HBM-bound: 520 B in, 72-136 B out per brick.
// the two bytes of one voxel (cluster.hpp:13-26 under -ffp-contract=off; 0xff: no leaf, cluster.hpp:28-31, 62-65)
__device__ __forceinline__ void cluster_bytes(float sum, uint32_t n, float trunc_recip, uint32_t* sd_bits, uint32_t* w_bits) {
    *sd_bits = 0xffu; *w_bits = 0xffu;
    if (n) {
        float sd = sum / float(n);                                      // the mean, as chad_map_export_leaves reports it
        sd = fminf(fmaxf(sd * trunc_recip, -1.0f), 1.0f);
        sd = sd * 127.0f + 127.0f;
        *sd_bits = uint32_t(sd);
        *w_bits = n < 254u ? n : 254u;
    }
}
// lane 8o + i holds byte i of word o of row `row`: byte row * 64 + lane of the words' array -- the wave stores the row's 64 bytes as
// one line (assembling the words in registers first took twelve cross-lane moves per voxel); out_mask[row]: which of the eight
// words hold a leaf (a voxel with observations has a distance byte below 0xff: such a word differs from kClusterEmpty)
__device__ __forceinline__ void cluster_store(uint32_t lane, size_t row, uint32_t sd_bits, uint32_t w_bits, bool leaf, uint64_t* out_tsdf, uint64_t* out_weight, uint8_t* out_mask) {
    reinterpret_cast<uint8_t*>(out_tsdf)[row * 64 + lane] = uint8_t(sd_bits);
    if (out_weight) reinterpret_cast<uint8_t*>(out_weight)[row * 64 + lane] = uint8_t(w_bits);
    if (out_mask) {
        const uint64_t leaves = __ballot(leaf);
        uint32_t mask = 0;
#pragma unroll
        for (uint32_t o = 0; o < 8; o++) mask |= ((leaves >> (8u * o)) & 0xffull) ? 1u << o : 0u;
        if (lane == 0) out_mask[row] = uint8_t(mask);
    }
}

__global__ void __launch_bounds__(256)
k_leaf_clusters(const uint32_t* __restrict__ order, uint32_t count, const uint64_t* __restrict__ brick_key,
                const float* __restrict__ brick_sum, const uint32_t* __restrict__ brick_cnt, float trunc_recip,
                uint64_t* __restrict__ out_key, uint64_t* __restrict__ out_tsdf, uint64_t* __restrict__ out_weight, uint8_t* __restrict__ out_mask) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t k = wave; k < count; k += waves) {
        const uint32_t b = order ? order[k] : k;
        const uint32_t n = brick_cnt[size_t(b) * 64 + lane];
        uint32_t sd_bits, w_bits;
        cluster_bytes(brick_sum[size_t(b) * 64 + lane], n, trunc_recip, &sd_bits, &w_bits);
        cluster_store(lane, k, sd_bits, w_bits, n != 0u, out_tsdf, out_weight, out_mask);
        if (lane == 0) out_key[k] = brick_key[b];
    }
}

// Walk the live bitmap: every wave takes kWalkWords words at a time (lane = word) and calls visit(slot) -- wave-uniformly,
// all 64 lanes active -- for each set bit.  Sixteen words = 512 slots per turn (64 until round 4): a wave works its bricks off one
// after the other (a handful of 512-byte rows in flight), so what hides the latency of those scattered rows is the NUMBER of waves,
// and a table of 2^23 slots now gives 16 384 turns to a grid of up to 4 096 workgroups instead of 4 096 turns to 512 (walk_grid).
constexpr uint32_t kWalkWords = 16;
// workgroups of four waves for a kernel that walks the bitmap of a table of hash_mask + 1 slots holding about `live` bricks (the
// host's mirror of the live count; ~0u = unknown): a turn per wave for a well-filled table, no more than 512 workgroups for a table
// with few bricks (a cfg2 frame's 9 k: the walk is then a scan of empty words, and launching 2 048 workgroups for it costs the
// headline's reset 1 %)
inline uint32_t walk_grid(uint32_t hash_mask, uint32_t live = ~0u) {
    const uint32_t turns = ((hash_mask >> 5) + kWalkWords) / kWalkWords;
    const uint32_t groups = (turns + 3u) / 4u;
    const uint32_t cap = live < (1u << 16) ? 512u : 4096u;
    return groups < 256u ? 256u : (groups > cap ? cap : groups);
}
template <class Visit>
__device__ __forceinline__ void for_each_live_brick(const MapView& m, Visit visit) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t words = (m.hash_mask >> 5) + 1u;
    for (uint32_t w0 = wave * kWalkWords; w0 < words; w0 += waves * kWalkWords) {
        const uint32_t word = lane < kWalkWords && w0 + lane < words ? m.live_bits[w0 + lane] : 0u;
        for (uint64_t holders = __ballot(word != 0u); holders; holders &= holders - 1ull) {
            const uint32_t l = uint32_t(__ffsll((unsigned long long)holders)) - 1u;
            for (uint32_t bits = uint32_t(__builtin_amdgcn_readlane(int(word), int(l))); bits; bits &= bits - 1u)
                visit((w0 + l) * 32u + uint32_t(__ffs(int(bits))) - 1u);
        }
    }
}

// Dense copy of the live bricks, in no particular order (the export sorts by key afterwards).  A wave reserves
// output rows for all bricks of its 64 bitmap words with ONE returning atomic (same-address atomics serialise at
// ~25 ns each on gfx950: one per brick would cost more than the copy).
__global__ void __launch_bounds__(256)
k_compact_bricks(MapView m, uint32_t* __restrict__ cursor, uint32_t capacity, uint64_t* __restrict__ out_key, float* __restrict__ out_sum,
                 uint32_t* __restrict__ out_cnt, float trunc) {
    const double metres_per_unit = double(trunc) / double(kSumOne);
    const bool pack_only = table_pack_only(m);   // (this kernel runs behind the table's last update)
    const uint32_t lane = threadIdx.x & 63;
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
        // four bricks at a time: a wave copies its bricks one after the other, and with one brick per turn that is a chain of
        // round trips -- 1.5 ms for the million blocks of a cfg5 submap, beside the frames of the next one
        uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0, queued = 0;
        auto copy_queued = [&]() {
            const uint32_t slot[4] = {q0, q1, q2, q3};
            long long units[4]; uint32_t count[4]; uint64_t key[4];
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if (i < queued && k + i < capacity) {
                    voxel_totals(m, size_t(slot[i]) * 64 + lane, &units[i], &count[i], pack_only);
                    if (lane == 0) key[i] = m.hash_keys[slot[i]];
                }
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if (i < queued && k + i < capacity) {
                    const float sum = float(double(units[i]) * metres_per_unit);   // exact integer -> one rounding
                    out_sum[size_t(k + i) * 64 + lane] = (count[i] & kPoisonedCount) ? __int_as_float(0x7fc00000) : sum;
                    out_cnt[size_t(k + i) * 64 + lane] = count[i] & ~kPoisonedCount;
                    if (lane == 0) out_key[k + i] = key[i];
                }
            k += queued;
            queued = 0;
        };
        for (; holders; holders &= holders - 1ull) {
            const uint32_t l = uint32_t(__ffsll((unsigned long long)holders)) - 1u;
            for (uint32_t bits = uint32_t(__builtin_amdgcn_readlane(int(word), int(l))); bits; bits &= bits - 1u) {
                const uint32_t slot = (w0 + l) * 32u + uint32_t(__ffs(int(bits))) - 1u;
                if (queued == 0) q0 = slot; else if (queued == 1) q1 = slot; else if (queued == 2) q2 = slot; else q3 = slot;
                if (++queued == 4) copy_queued();
            }
        }
        if (queued) copy_queued();
    }
}

// CHAD_FLAG_DAG_ONLY: a closed submap is its DAG and nothing else, so its bricks go from the table straight to cluster words -- what
// k_compact_bricks (f32 sum with one rounding, count) followed by k_leaf_clusters would deliver, bit for bit, without the dense copy in
// between (0.85 GB written and read again per cfg5 submap).  The fold wants the blocks in key order, so the ORDER comes first: the
// live slots' keys and slot numbers (k_table_keys: 12 bytes per block, in no particular order), a sort of the keys with their
// position as payload, and only then the bricks, read once in that order, each cluster row written where it belongs
// (k_sorted_clusters).  (Cluster words in table order first, then sorted and gathered: 0.53 + 0.11 ms instead of 0.03 + 0.25.)
__global__ void __launch_bounds__(256)
k_table_keys(MapView m, uint32_t* __restrict__ cursor, uint32_t capacity, uint64_t* __restrict__ out_key, uint32_t* __restrict__ out_slot) {
    __shared__ uint32_t wave_total[4], group_base;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t words = (m.hash_mask >> 5) + 1u;
    // 256 bitmap words = 8 192 slots per workgroup step and ONE returning atomic on the cursor for them (per wave and sixteen words
    // it was 8 192 atomics for a table of 2^22 slots, queueing on one address for 25 ns each: 0.2 ms)
    for (uint32_t g0 = blockIdx.x * 256u; g0 < words; g0 += gridDim.x * 256u) {
        const uint32_t w = g0 + threadIdx.x;
        const uint32_t word = w < words ? m.live_bits[w] : 0u;
        uint32_t mine = uint32_t(__popc(word)), before = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(before, o); if (int(lane) >= o) before += t; }
        __syncthreads();   // (the previous step's readers of group_base are through)
        if (lane == 63u) wave_total[wave] = before;
        __syncthreads();
        if (threadIdx.x == 0) { const uint32_t all = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3]; group_base = all ? atomicAdd(cursor, all) : 0u; }
        __syncthreads();
        uint32_t k = group_base + before - mine;
        for (uint32_t v = 0; v < wave; v++) k += wave_total[v];
        // the words' bits, a byte of the bitmap per lane and turn (a word per lane would leave a lane up to 32 dependent rounds)
        for (uint32_t part = 0; part < 4; part++) {
            const uint32_t src = 16u * part + (lane >> 2);   // lane of this wave that holds the word
            const uint32_t src_word = uint32_t(__shfl(int(word), int(src))), src_k = uint32_t(__shfl(int(k), int(src)));
            const uint32_t shift = 8u * (lane & 3u);
            uint32_t bits = (src_word >> shift) & 0xffu;
            uint32_t at = src_k + uint32_t(__popc(src_word & ((1u << shift) - 1u)));
            for (; bits; bits &= bits - 1u, at++) {
                const uint32_t slot = (g0 + 64u * wave + src) * 32u + shift + uint32_t(__ffs(int(bits))) - 1u;
                if (at < capacity) { out_key[at] = m.hash_keys[slot]; out_slot[at] = slot; }
            }
        }
    }
}
// rank r of the sorted keys is entry order[r] of k_table_keys' list: its brick -> the eight cluster words of row r (a wave per brick, a lane per voxel)
__global__ void __launch_bounds__(256)
k_sorted_clusters(MapView m, const uint32_t* __restrict__ order, const uint32_t* __restrict__ slots, uint32_t count, float trunc, float trunc_recip,
                  uint64_t* __restrict__ out_tsdf, uint64_t* __restrict__ out_weight, uint8_t* __restrict__ out_mask) {
    const double metres_per_unit = double(trunc) / double(kSumOne);
    const bool pack_only = table_pack_only(m);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    uint32_t ahead[4];   // the slots of the step after this one: two dependent loads that would otherwise stand in front of every step's bricks
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) ahead[i] = wave * 4u + i < count ? slots[order[wave * 4u + i]] : 0u;
    for (uint32_t r0 = wave * 4u; r0 < count; r0 += waves * 4u) {   // four bricks' loads in flight together
        uint32_t slot[4];
        long long units[4]; uint32_t cnt[4];
        const uint32_t r1 = r0 + waves * 4u;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) { slot[i] = ahead[i]; ahead[i] = r1 + i < count ? slots[order[r1 + i]] : 0u; }
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) if (r0 + i < count) voxel_totals(m, size_t(slot[i]) * 64 + lane, &units[i], &cnt[i], pack_only);
#pragma unroll
        for (uint32_t i = 0; i < 4; i++)
            if (r0 + i < count) {
                const float sum = (cnt[i] & kPoisonedCount) ? __int_as_float(0x7fc00000) : float(double(units[i]) * metres_per_unit);   // k_compact_bricks' archive value
                uint32_t sd_bits, w_bits;
                cluster_bytes(sum, cnt[i] & ~kPoisonedCount, trunc_recip, &sd_bits, &w_bits);
                cluster_store(lane, size_t(r0 + i), sd_bits, w_bits, (cnt[i] & ~kPoisonedCount) != 0u, out_tsdf, out_weight, out_mask);
            }
    }
}

// Octree::clear() (octree.hpp:22-29) in O(H/32 + bricks) and without a host round trip: every live brick's key is
// released and its rows are zeroed; then the bitmap words and the counter stripes.
__global__ void __launch_bounds__(256)
k_clear_bricks(MapView m) {
    const uint32_t lane = threadIdx.x & 63;
    const bool pack_only = table_pack_only(m);   // exact rows never written since the last clear: they are zero already
    for_each_live_brick(m, [&](uint32_t slot) {
        if (lane == 0) m.hash_keys[slot] = kEmptyKey;
        if (!pack_only) {
            m.brick_sum[size_t(slot) * 64 + lane] = 0ll;
            m.brick_cnt[size_t(slot) * 64 + lane] = 0u;
        }
        m.brick_pack[size_t(slot) * 64 + lane] = 0ull;
    });
    // each wave clears exactly the words it has just read
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t words = (m.hash_mask >> 5) + 1u;
    for (uint32_t w0 = wave * kWalkWords; w0 < words; w0 += waves * kWalkWords)
        if (lane < kWalkWords && w0 + lane < words && m.live_bits[w0 + lane]) m.live_bits[w0 + lane] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < kCountStripes) m.brick_count[threadIdx.x * 16] = 0u;
    // (the table's flag word is NOT touched here: every workgroup reads its "exact rows" bit above, at its own time.  It is reset
    //  behind this kernel -- k_table_flags, or the memset of a reset -- before the table takes updates again: chad_hip.hip, table_cleared)
}
// the flag word of a table that has just been cleared: `keep` = the bits that survive (kTableOverflowed is sticky until a reset)
__global__ void k_table_flags(uint32_t* flags, uint32_t keep) { atomicAnd(flags, keep); }

// Move the bricks of a map into a larger, empty one (capacity growth between frames).
__global__ void __launch_bounds__(256)
k_rehash_bricks(MapView from, MapView to) {
    const uint32_t lane = threadIdx.x & 63;
    const bool pack_only = table_pack_only(from);   // (the new table's exact rows are zero from its allocation; the host copies the flag word over)
    for_each_live_brick(from, [&](uint32_t slot) {
        uint32_t dst = 0;
        if (lane == 0) dst = find_or_create_brick(to, from.hash_keys[slot]);
        dst = uint32_t(__builtin_amdgcn_readfirstlane(int(dst)));
        if (dst == kNoBrick) return;
        if (!pack_only) {
            to.brick_sum[size_t(dst) * 64 + lane] = from.brick_sum[size_t(slot) * 64 + lane];
            to.brick_cnt[size_t(dst) * 64 + lane] = from.brick_cnt[size_t(slot) * 64 + lane];
        }
        to.brick_pack[size_t(dst) * 64 + lane] = from.brick_pack[size_t(slot) * 64 + lane];
    });
}

// Move what the scatter form has staged into the exact sums (before the staging word's count field could run out).  Safe beside
// updates in flight: the word is taken with an exchange and added with atomics, so nothing is lost or counted twice, and every
// reader adds the two parts anyway.
__global__ void __launch_bounds__(256)
k_fold_packed(MapView m) {
    const uint32_t lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) mark_exact_rows(m);
    for_each_live_brick(m, [&](uint32_t slot) {
        const size_t at = size_t(slot) * 64 + lane;
        if (m.brick_pack[at] == 0ull) return;
        const unsigned long long word = atomicExch(&m.brick_pack[at], 0ull);
        if (!word) return;
        long long units; uint32_t count;
        unpack_staged(word, &units, &count);
        atomicAdd(reinterpret_cast<unsigned long long*>(&m.brick_sum[at]), (unsigned long long)units);
        atomicAdd(&m.brick_cnt[at], count);
    });
}

// keys for sorting bricks by block key: key' = brick key (ascending), payload = brick index (implicit)
__global__ void k_copy_u64(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, uint32_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) dst[i] = src[i];
}

}  // namespace chadhip

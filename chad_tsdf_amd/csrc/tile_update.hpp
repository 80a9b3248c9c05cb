// tile_update.hpp -- the TSDF update with the voxel neighbourhood staged in LDS (reference: octree.hpp:80-170).
//
// Points arrive grouped by "super block": 2x2x2 point blocks = 8x8x8 voxels (the sort key of frame_kernels.hpp puts
// them next to each other).  A ray segment reaches at most R = ceil(trunc/res) + 1 voxels away from its point, so
// every voxel a super block's rays touch lies in the cube of D = 8 + 2R voxels around it: D^3 x 8 bytes of LDS --
// 21.4 KiB for trunc = 2*res (D = 14), 32 KiB for D = 16 (the largest tile kept in LDS) -- which the 4x4x4 = 64
// bricks around the super block cover.  One workgroup of four waves per work item (a super block, cut at every
// kItemPoints-th = 1024th point by k_frame_gather); the first version staged one 4x4x4 block per wave: 3.2 times as many items,
// each paying the per-item latencies below, and 83 % of the lanes busy instead of 97 %.  Per item:
//   1. the waves take the item's 64-point chunks in turn; each lane walks the ray of its point and accumulates
//      (d, 1) into the tile with ONE ds_add_u64 per observation.  A tile cell is a 64-bit word
//      [count : 20 | sum of (d * 2^30/trunc + 2^30) : 44]: the fixed-point step trunc * 2^-30 is finer than the f32
//      rounding of d itself; an item holds at most kItemPoints = 1024 points, each of which visits a cell at most once,
//      so a cell takes at most 1024 observations of at most 2^31 each: 2^41 < 2^44, no carry into the count, and the
//      count (<= 1024) fits its 20 bits; integer sums do not depend on the order of arrival.
//      The walk itself is the reference's traversal (same f32 operations on tmax/delta, same comparison tree, so
//      the voxel lists are bit-identical) with the bookkeeping folded into three registers: the LDS address and the
//      per-axis remaining step counts share one word that a step updates with a single add, and the signed
//      distance -- for which the bar is 1e-4, not bit equality -- is carried incrementally in fixed-point scale.
//      ~20 VALU instructions per voxel step (the first version: ~110).
//      Rays that leave the tile (float rounding at the rim, trunc > 3*res) or whose normal is not finite (Q4) take
//      the global path of k_update_scatter instead, outside the hot loop;
//   2. workgroup barrier; wave w owns the 16 bricks of slab z = w from here on.  It reads them back -- sixteen lanes per brick,
//      a row of four cells each, four bricks per pass, all reads in flight together -- to learn which received observations;
//      lanes 0..15 whose brick was touched claim or find its slot with one CAS (map_kernels.hpp);
//   3. every touched brick is flushed with two wave-wide atomic adds (512 B of 64-bit sums, 256 B of counts), the full-rate shape for
//      memory-side atomics, and its part of the tile is zeroed for the next item; workgroup barrier.
// Memory operations of a wave complete in issue order, so a load issued behind the flush's atomics waits ~3000
// cycles for them.  The loop is therefore software-pipelined: right after the first barrier -- before the flush --
// every wave loads the NEXT item's first chunk of points, and the next item's record itself was fetched by wave 0 during the walk.
// (Until round 4 it also loaded the home hash slots of its 16 bricks there, so that a brick whose key was already in place needed
// no CAS: 64 look-ups per item whatever it touches -- 146 k random 128-byte reads per 1 M-point frame, 18.7 of the kernel's 51.4 MB
// read, rocprofv3 request counters with and without -- to save CASes that are in flight behind the next item's loads anyway:
// without them the kernel alone takes 27.8 instead of 29.2 us and the frame rate is 2.3 % higher, scripts/ab_bench.sh.)
// Work items: the first one is static (workgroup index; its record and the lists' stripe sums come ready-made from
// k_frame_normals' last workgroup, FrameHeader::first_item / item_first: one round trip in front of the first ray instead of
// three), the rest are pulled from 256 separate counters (a returning atomic on ONE shared address serialises at ~25 ns per
// dequeue).  Sparse frames (few points per block): in
// automatic mode the kernel notices (device memory) and runs the scatter form (map_kernels.hpp, scatter_update) on its
// own grid instead -- with one point per block the staging is pure overhead.
#pragma once
#include <hip/hip_ext.h>

#include "map_kernels.hpp"

namespace chadhip {

constexpr int kTileWaves = 4;            // waves per workgroup: they share the item's tile and take its chunks in turn
constexpr int kTileThreads = 64 * kTileWaves;
constexpr uint32_t kMaxReach = 4;        // D = 16: 32 KiB of LDS per workgroup
constexpr float kFixedOne = kSumOne;  // 2^30: |d| = trunc in the tile's fixed-point scale -- the unit of the map's sums (map_kernels.hpp)
constexpr uint32_t kCountShift = 44;        // tile cell = [count : 20 | sum of biased fixed-point distances : 44]: an item's kItemPoints = 1024
                                            // observations of at most 2^31 each stay below 2^41 (with the count at bit 40 a cell that took 512
                                            // observations of +trunc carried into its own count: found by test_sort_survives_..., whose small
                                            // frames put thousands of points into the voxels next to the sensor)
static_assert(uint64_t(kItemPoints) << 31 < (1ull << kCountShift) && kItemPoints < (1u << (64 - kCountShift)), "a tile cell must hold an item's worst case");

// b where the lane's bit of `mask` is set, a elsewhere: v_cndmask_b32 with the lane mask in a scalar register pair (the inverse of a
// ballot: the compiler takes the 64-bit mask as the select's condition as it stands.  Until round 5 these were inline asm, behind each
// of which the compiler put an s_nop -- six per voxel step, an issue slot each)
__device__ __forceinline__ float select_f(float a, float b, uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask) ? b : a; }
__device__ __forceinline__ uint32_t select_u(uint32_t a, uint32_t b, uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask) ? b : a; }

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

// LDS-only workgroup barrier: __syncthreads() would also drain the global atomics in flight.
__device__ __forceinline__ void tile_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool kStamps, int kMinWaves>
__global__ void __launch_bounds__(kTileThreads, kMinWaves)
k_update_tiled(const float4* __restrict__ points0, const float4* __restrict__ points1, const float4* __restrict__ normals, const uint4* __restrict__ items,
               FrameHeader* __restrict__ h, F3 pos, float res, float recip, float trunc, MapView m, uint32_t reach, uint32_t adaptive, uint32_t items_capacity /* entries per stripe of items[] */,
               unsigned long long* stamps, uint32_t* pack_counter, uint32_t* pack_seen_host) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
    unsigned long long wall_entry = 0, wall_loop = 0, wall_zeroed = 0, wall_lists = 0;   // kStamps: constant-clock (10 ns) times of the wave's entry and of its first item
    if (kStamps) wall_entry = wall_clock64();
    const float4* __restrict__ pts = (h->sort_passes & 1u) ? points0 : points1;   // the frame in processing order: k_frame_normals' output (the sort's other buffer)
    const int R = int(reach), D = 8 + 2 * R;
    const uint32_t cells = uint32_t(D * D * D);
    const uint32_t lane = threadIdx.x & 63u, wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
    // cells + 2 words: the last two are never added to -- cell `cells` stands in for cells outside the tile, and the pair is what
    // the read-back of step 2 looks at for rows outside the tile; then the next item's record
    unsigned long long* tile = reinterpret_cast<unsigned long long*>(tile_raw);
    uint4* next_record = reinterpret_cast<uint4*>(tile + ((cells + 3u) & ~1u));
    const uint32_t tile_base = uint32_t(uintptr_t(tile));  // LDS byte address (low half of the flat address)
    const float to_fixed = kFixedOne / trunc;
    // Everything the workgroup needs to know about the frame's work lists comes ready-made from k_frame_normals (FrameHeader::
    // item_first, frame_sparse, first_item): ONE round trip, in flight while the tile is zeroed -- the long items of all stripes
    // first, then the short ones (k_frame_gather); where each stripe's entries begin goes to LDS (registers are scarce here):
    // first_long[s], first_short[s], s = 0 .. kListStripes.
    uint32_t* first_long = reinterpret_cast<uint32_t*>(next_record + 1);
    uint32_t* first_short = first_long + kListStripes + 1;
    uint32_t summary_word = 0;
    if (threadIdx.x < 2 * (kListStripes + 1)) summary_word = (&h->item_first[0][0])[threadIdx.x];
    const uint32_t sparse = adaptive ? h->frame_sparse : 0u;
    uint4 record = h->first_item[blockIdx.x];   // (meaningless beyond the frame's last item: masked below)
    for (uint32_t k = threadIdx.x; k <= cells + 1u; k += kTileThreads) tile[k] = 0ull;
    if (kStamps) wall_zeroed = wall_clock64();
    if (threadIdx.x < 2 * (kListStripes + 1)) first_long[threadIdx.x] = summary_word;
    tile_barrier();   // the tile is zeroed, the lists are in LDS
    if (sparse) {  // about one point per block: staging tiles is pure overhead, take the general form
        // (its notes go where the tile would be: as many rows of kStageRow entries per wave as that leaves room for -- ten for a
        // reach of 3 voxels, fifteen for 4: a ray's steps grow with the reach as the tile does)
        const uint32_t stage_rows = ((cells + 2u) / kTileWaves) / kStageRow;
        scatter_update(pts, normals, h, pos, res, recip, trunc, m, pack_counter, pack_seen_host, tile + wave * (stage_rows * kStageRow), stage_rows);
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) mark_exact_rows(m);   // the tile flush adds to the exact rows (map_kernels.hpp, kTableExactRows)
    if (kStamps) wall_lists = wall_clock64();
    // (wave-uniform: kept in scalar registers -- as vector registers they were the two the register allocator spilled to scratch)
    const uint32_t n_long = uint32_t(__builtin_amdgcn_readfirstlane(int(first_long[kListStripes])));
    const uint32_t nitems = n_long + uint32_t(__builtin_amdgcn_readfirstlane(int(first_short[kListStripes])));
    auto item_record = [&](uint32_t v) {
        const bool is_long = v < n_long;
        const uint32_t* first = is_long ? first_long : first_short;
        const uint32_t i = is_long ? v : v - n_long;
        uint32_t stripe = 0;
#pragma unroll
        for (uint32_t s = 1; s < kListStripes; s++) stripe = i >= first[s] ? s : stripe;
        const uint32_t index = i - first[stripe];
        return items[is_long ? stripe * items_capacity + index : (stripe + 1u) * items_capacity - 1u - index];
    };
    uint32_t applied = 0;

    const uint32_t group = blockIdx.x % kWorkGroups;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t_begin = 0, items_done = 0;
    if (kStamps) { t_begin = stamp_now(); }

    // everything a wave holds about an item before it starts on it
    if (blockIdx.x >= nitems) record = make_uint4(0u, 0u, 0u, 0u);   // length 0: no more work
    float4 q_first = make_float4(0, 0, 0, 0), n_first = make_float4(0, 0, 0, 0);            // its first chunk's point and normal for this lane
    auto prefetch = [&](const uint4& rec) {
        const uint32_t first = uint32_t(__builtin_amdgcn_readfirstlane(int(rec.x))), len = uint32_t(__builtin_amdgcn_readfirstlane(int(rec.y)));
        const uint32_t k = lane * ((len + 63u) >> 6) + wave;   // chunk `wave` of the item (see the walk)
        if (k < len) { q_first = pts[first + k]; n_first = normals[first + k]; }
    };
    // key of brick (bx, by, wave) = lane of the super block at `code`: the corner's code with -4 / 0 / +4 / +8 voxels added per axis, in
    // interleaved ("dilated") arithmetic -- a subtraction borrows through the zero gaps of the masked operand, an addition carries
    // through gaps filled with ones; the corner is a multiple of 8, so +4 just sets a bit.  The four candidates per axis are
    // wave-uniform (scalar unit); a lane only selects.  (Worked out behind the walk, from the code in scalar registers: held in vector
    // registers across the walk -- as until round 5 -- it is three of the registers that stand between this kernel and a sixth wave per SIMD.)
    auto brick_key = [&](uint64_t code, uint32_t l) {
        constexpr uint64_t kMx = 0x1249249249249249ull, kMy = kMx << 1, kMz = kMx << 2;
        const uint64_t xd = code & kMx, yd = code & kMy, zd = code & kMz;
        const uint64_t x0 = (xd - (1ull << 6)) & kMx, x2 = xd | (1ull << 6), x3 = ((xd | ~kMx) + (1ull << 9)) & kMx;
        const uint64_t y0 = (yd - (1ull << 7)) & kMy, y2 = yd | (1ull << 7), y3 = ((yd | ~kMy) + (1ull << 10)) & kMy;
        const uint64_t zs = wave == 0u ? (zd - (1ull << 8)) & kMz : wave == 1u ? zd : wave == 2u ? zd | (1ull << 8) : ((zd | ~kMz) + (1ull << 11)) & kMz;
        const uint32_t bx = l & 3u, by = (l >> 2) & 3u;
        return (bx == 0u ? x0 : bx == 1u ? xd : bx == 2u ? x2 : x3) | (by == 0u ? y0 : by == 1u ? yd : by == 2u ? y2 : y3) | zs;
    };
    prefetch(record);
    uint32_t turn = 0;  // first lane of the last wave: ticket of the item after the current one (in flight during the walk)
    tile_barrier();     // the tile is zeroed
    if (kStamps) wall_loop = wall_clock64();

    while (__builtin_amdgcn_readfirstlane(int(record.y)) != 0) {
        if (kStamps) { t0 = stamp_now(); items_done++; }
        const uint32_t first = uint32_t(__builtin_amdgcn_readfirstlane(int(record.x))), len = uint32_t(__builtin_amdgcn_readfirstlane(int(record.y)));
        const uint64_t corner_code = uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(record.z)))) | (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(record.w)))) << 32);
        const I3 corner = morton_decode(corner_code);                         // lowest voxel of the super block
        const int ox = corner.x - R, oy = corner.y - R, oz = corner.z - R;    // tile origin
        if (threadIdx.x == kTileThreads - 64) turn = atomicAdd(&h->work_cursor[group * 16u], 1u);
        if (kStamps) { const unsigned long long t1 = stamp_now(); ph[0] += t1 - t0; t0 = t1; }

        // step 1: the walk (the wave's first chunk is already in registers)
        float4 q = q_first, nq = n_first;
        // Chunk c of an item with C chunks holds the points lane * C + c: neighbours in processing order -- points of one
        // voxel, whose rays visit the same cells in the same steps and serialise in the LDS atomic units -- are dealt to
        // different chunks instead of neighbouring lanes.  (The loads become strided; they are prefetched.)
        const uint32_t chunks = (len + 63u) >> 6;
        for (uint32_t c = wave; c < chunks; c += kTileWaves) {
            const uint32_t k = lane * chunks + c;
            float4 q_next = q, n_next = nq;
            if (k < len) {
                const F3 p{q.x, q.y, q.z}, nrm{nq.x, nq.y, nq.z};
                RayWalk w;
                w.begin(p, pos, res, recip, trunc);
                const int lx = w.cur.x - ox, ly = w.cur.y - oy, lz = w.cur.z - oz;
                const uint32_t spx = uint32_t(abs(w.last.x - w.cur.x)), spy = uint32_t(abs(w.last.y - w.cur.y)), spz = uint32_t(abs(w.last.z - w.cur.z));
                // every voxel of the walk lies in the box spanned by its first and last voxel
                const bool inside = uint32_t(lx) < uint32_t(D) && uint32_t(ly) < uint32_t(D) && uint32_t(lz) < uint32_t(D) &&
                                    uint32_t(w.last.x - ox) < uint32_t(D) && uint32_t(w.last.y - oy) < uint32_t(D) && uint32_t(w.last.z - oz) < uint32_t(D) &&
                                    (spx | spy | spz) < 8u && finite3(nrm);
                // the wave's next chunk: loads issued here, behind the register-hungry traversal setup, in flight during the steps
                if (c + kTileWaves < chunks && k + kTileWaves < len) { q_next = pts[first + k + kTileWaves]; n_next = normals[first + k + kTileWaves]; }
                if (inside) {
                    LeanRay r;
                    r.tx = w.tmax.x; r.ty = w.tmax.y; r.tz = w.tmax.z;
                    r.dx = w.delta.x; r.dy = w.delta.y; r.dz = w.delta.z;
                    asm("" : "+v"(r.dx), "+v"(r.dy), "+v"(r.dz));   // |res * inv| as values: keeps the fabs out of the step loop
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
                        __hip_atomic_fetch_add(lds_cell(r.s >> 16), (1ull << kCountShift) | fixed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        // comparison tree of octree.hpp:126-150 as lane masks in scalar registers: x steps where tx < ty and
                        // tx < tz, y where not (tx < ty) and ty < tz, z elsewhere.  The selects take the mask as their SGPR
                        // operand directly (hipcc turns "neither x nor y" into a five-instruction detour through VGPRs)
                        const uint64_t bxy = __ballot(r.tx < r.ty), bxz = __ballot(r.tx < r.tz), byz = __ballot(r.ty < r.tz);
                        const uint64_t mx = bxy & bxz, my = ~bxy & byz, mz = ~(mx | my);
                        r.tx += select_f(0.0f, r.dx, mx);   // tmax >= 0 (or NaN/inf): adding +0.0f leaves it unchanged
                        r.ty += select_f(0.0f, r.dy, my);
                        r.tz += select_f(0.0f, r.dz, mz);
                        r.d += select_f(select_f(r.ddz, r.ddy, my), r.ddx, mx);
                        r.s += select_u(select_u(r.kz, r.ky, my), r.kx, mx);
                    } while ((r.s & 0x888u) == 0x888u);
                } else {  // straight to HBM, one observation at a time (rare)
                    uint64_t far_key = kEmptyKey;
                    uint32_t far_brick = kNoBrick;
                    do {
                        const uint64_t mc = morton_encode(w.cur.x, w.cur.y, w.cur.z);
                        const uint64_t key = mc & kBlockMask;
                        if (key != far_key) { far_brick = find_or_create_brick(m, key); far_key = key; }
                        if (far_brick != kNoBrick) {
                            const size_t at = size_t(far_brick) * 64 + (uint32_t(mc) & 63u);
                            add_observation(m, at, observe(w.cur, p, nrm, res, trunc), to_fixed);
                            applied++;
                        }
                    } while (w.next());
                }
            }
            q = q_next; nq = n_next;
        }
        // the last wave -- the one with the fewest chunks -- fetches the next item's record while the others finish walking:
        // ticket t -> item gridDim + group + 256 t
        if (threadIdx.x == kTileThreads - 64) {
            const uint32_t next = gridDim.x + group + kWorkGroups * turn;
            *next_record = next < nitems ? item_record(next) : make_uint4(0u, 0u, 0u, 0u);
        }
        if (kStamps) { const unsigned long long t1 = stamp_now(); ph[3] += t1 - t0; t0 = t1; }
        tile_barrier();  // all LDS adds of this item have been performed; the next record is in place
        if (kStamps) { const unsigned long long t1 = stamp_now(); ph[1] += t1 - t0; t0 = t1; }

        // step 2: which of this wave's 16 bricks received observations (four reads in flight at a time)
        uint32_t lane_here = lane;
        asm volatile("" : "+v"(lane_here));   // (what follows is worked out per item, behind the walk: hoisted out of the loop it would live in registers across it)
        const uint64_t this_key = lane_here < 16u ? brick_key(corner_code, lane_here) : kEmptyKey;
        const uint32_t this_slot = hash_block(this_key, m.hash_mask);
        // Only "does any cell of the brick hold something" is asked here, so the lanes need not sit on their voxels: 16 lanes per
        // brick, one per (y, z) row of four cells along x, four bricks (bx = 0..3 of one by) per pass -- 4 passes instead of 16, no
        // branches, all reads in flight together (the voxel-per-lane form compiled to sixteen dependent LDS round trips under
        // exec masks: 3 300 cycles per item).  A row outside the tile reads the two spare zero cells; a brick cut by the tile's
        // rim along x reads its cells inside (two at least, for reach >= 2), some of them twice.
        uint32_t touched = 0;
        {
            const int rx = 4 * int(lane_here >> 4) + R - 4, ry = int(lane_here & 3u) + R - 4, rz = 4 * int(wave) + int((lane_here >> 2) & 3u) + R - 4;
            const int lo = max(rx, 0), hi = min(rx + 3, D - 1);
            const uint32_t zero_at = tile_base + 8u * cells;
            const uint32_t row_a = tile_base + 8u * uint32_t(lo + D * ry + D * D * rz), delta_b = 8u * uint32_t(max(hi - 1 - lo, 0));
            const bool z_ok = uint32_t(rz) < uint32_t(D);
            unsigned long long any[4];
#pragma unroll
            for (int by = 0; by < 4; by++) {
                const bool ok = z_ok && uint32_t(ry + 4 * by) < uint32_t(D);
                const uint32_t a = ok ? row_a + uint32_t(by) * uint32_t(32 * D) : zero_at;
                const uint32_t b = ok ? a + delta_b : zero_at;
                any[by] = lds_cell(a)[0] | lds_cell(a)[1] | lds_cell(b)[0] | lds_cell(b)[1];
            }
#pragma unroll
            for (int by = 0; by < 4; by++) {
                const uint64_t rows = __ballot(any[by] != 0ull);
                const uint32_t t4 = ((rows & 0xffffull) ? 1u : 0u) | ((rows & 0xffff0000ull) ? 2u : 0u) | ((rows & 0xffff00000000ull) ? 4u : 0u) | ((rows >> 48) ? 8u : 0u);
                touched |= t4 << (4 * by);
            }
        }
        if (kStamps) { asm volatile("" :: "s"(touched)); const unsigned long long t1 = stamp_now(); ph[6] += t1 - t0; t0 = t1; }
        // touched bricks: one CAS claims or finds the home slot.  It is
        // issued here and resolved behind the next item's loads (below), so its round trip overlaps their address arithmetic
        const bool mine = lane < 16u && ((touched >> lane) & 1u);
        const bool ask = mine;
        uint64_t seen = this_key;
        if (ask) seen = atomicCAS((unsigned long long*)&m.hash_keys[this_slot], (unsigned long long)kEmptyKey, (unsigned long long)this_key);
        // the loads for the next item (in flight during the rest of step 2 and step 3)
        record = *next_record;
        prefetch(record);
        uint32_t my_brick = mine ? this_slot : kNoBrick;
        if (ask && seen != this_key) {
            if (seen == kEmptyKey) {  // claimed (map_kernels.hpp, find_or_create_brick)
                atomicOr(&m.live_bits[this_slot >> 5], 1u << (this_slot & 31u));
                atomicAdd(&m.brick_count[(this_slot & (kCountStripes - 1)) * 16], 1u);
            } else my_brick = find_or_create_brick(m, this_key, (this_slot + 1u) & m.hash_mask);  // the home slot holds another block
        }
        // the prefetched loads are older than the flush's atomics: wait for them here, where they have long arrived,
        // and not at their first use, where the in-order counter would make that wait cover the atomics as well
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(q_first.x), "+v"(q_first.y), "+v"(q_first.z), "+v"(n_first.x), "+v"(n_first.y), "+v"(n_first.z));
        if (kStamps) { asm volatile("" :: "v"(my_brick)); const unsigned long long t1 = stamp_now(); ph[7] += t1 - t0; t0 = t1; }

        // step 3: flush.  lane = voxel v of a brick (Morton order of the low 6 code bits).  This wave's bricks: (bx, by) = 0..3 x 0..3 in
        // slab bz = wave; brick b sits at voxel offset 4 * (b - 1) from the super block's corner.
        const int vx = int((lane_here & 1u) | ((lane_here >> 2) & 2u)), vy = int(((lane_here >> 1) & 1u) | ((lane_here >> 3) & 2u)), vz = int(((lane_here >> 2) & 1u) | ((lane_here >> 4) & 2u));
        const int tz = 4 * int(wave) + vz + R - 4;
        const int slab_cell = (vx + R - 4) + D * (vy + R - 4) + D * D * tz;   // this lane's cell in brick (0, 0) of the slab
        for (uint32_t left = touched; left;) {
            const int b = __ffs(int(left)) - 1;
            left &= left - 1u;
            const uint32_t brick = uint32_t(__builtin_amdgcn_readlane(int(my_brick), b));
            // (a voxel outside the tile reads the spare zero cell)
            const int tx = 4 * (b & 3) + vx + R - 4, ty = 4 * (b >> 2) + vy + R - 4;
            const bool inside_tile = uint32_t(tx) < uint32_t(D) && uint32_t(ty) < uint32_t(D) && uint32_t(tz) < uint32_t(D);
            const uint32_t at = inside_tile ? uint32_t(slab_cell + 4 * ((b & 3) + D * (b >> 2))) : cells;
            const unsigned long long cell = tile[at];
            if (cell) {
                const uint32_t c = uint32_t(cell >> kCountShift);
                // sum of d in units of trunc / 2^30 = sum of biased fixed values - c * 2^30: an exact integer, added as such
                const long long fixed_sum = (long long)(cell & ((1ull << kCountShift) - 1)) - ((long long)c << 30);
                if (brick != kNoBrick) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(&m.brick_sum[size_t(brick) * 64 + lane]), (unsigned long long)fixed_sum);
                    atomicAdd(&m.brick_cnt[size_t(brick) * 64 + lane], c);
                    applied += c;
                }
                tile[at] = 0ull;
            }
        }
        if (kStamps) { const unsigned long long t1 = stamp_now(); ph[4] += t1 - t0; t0 = t1; }
        tile_barrier();  // the tile is zero again
        if (kStamps) { const unsigned long long t1 = stamp_now(); ph[2] += t1 - t0; t0 = t1; }
    }
    if (kStamps && lane == 0) {
        const uint32_t wave_global = blockIdx.x * kTileWaves + wave;
        ph[5] = stamp_now() - t_begin;
        ph[1] += ph[2];         // both barriers
        ph[2] = items_done;
        for (int k = 0; k < 8; k++) stamps[size_t(wave_global) * 16 + k] = ph[k];
        stamps[size_t(wave_global) * 16 + 8] = wall_entry; stamps[size_t(wave_global) * 16 + 9] = wall_loop; stamps[size_t(wave_global) * 16 + 10] = wall_clock64();
        stamps[size_t(wave_global) * 16 + 11] = wall_zeroed; stamps[size_t(wave_global) * 16 + 12] = wall_lists;
    }
    unsigned long long total = applied;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    if (lane == 0 && total) atomicAdd(&h->n_updates[((blockIdx.x * kTileWaves + wave) % kUpdateStripes) * 16u], total);
}

inline uint32_t tile_reach(float res, float trunc) {
    const float r = std::ceil(trunc / res) + 1.0f;
    return r < 2.0f ? 2u : (r > float(kMaxReach) ? kMaxReach : uint32_t(r));   // (at least 2: the read-back wants two cells of every brick inside the tile along x)
}

// Launch shape of the update kernel for one map: fixed by its resolution and truncation distance (the tile's reach) and by the
// device it lives on -- computed once per map (chad_map_create), on the map's device: nothing here is process-wide, so maps on
// different devices, or created from different threads, do not share state.
struct TileLaunch {
    uint32_t reach = 0, lds = 0;
    uint32_t groups = 0, groups_stamped = 0;   // k_update_tiled: persistent workgroups, as many as are resident at once (registers and LDS decide), on every CU
};
inline TileLaunch tile_launch_shape(float res, float trunc, uint32_t cu_count) {
    TileLaunch t;
    t.reach = tile_reach(res, trunc);
    const uint32_t d = 8 + 2 * t.reach;
    t.lds = uint32_t(((size_t(d) * d * d + 3) & ~size_t(1)) * 8 + 16 + 2 * (kListStripes + 1) * 4 + 8);   // tile + two spare zero cells (padded to 16 bytes) + next item's record + list index
    int resident[2] = {1, 1};
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident[0], k_update_tiled<false, 6>, kTileThreads, t.lds);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident[1], k_update_tiled<true, 6>, kTileThreads, t.lds);
    (void)hipGetLastError();
    uint32_t per_cu[2];
    for (int k = 0; k < 2; k++) per_cu[k] = uint32_t(std::max(1, std::min(8, resident[k])));
    if (const char* e = std::getenv("CHAD_DEBUG_WAVES_PER_CU")) per_cu[0] = per_cu[1] = uint32_t(std::max(1, std::min(8, std::atoi(e))));
    // every workgroup finds its first item in FrameHeader::first_item (kFirstItems entries)
    t.groups = std::min<uint32_t>(cu_count * per_cu[0], kFirstItems);
    t.groups_stamped = std::min<uint32_t>(cu_count * per_cu[1], kFirstItems);

    return t;
}

inline void tile_update_enqueue(const TileLaunch& shape, const float4* points0, const float4* points1, const float4* normals, const uint4* items, FrameHeader* h, F3 pos, float res, float recip, float trunc,
                                const MapView& m, hipStream_t stream, bool adaptive, uint32_t items_capacity, uint32_t* pack_counter, uint32_t* pack_seen_host,
                                unsigned long long* stamps = nullptr, hipEvent_t started = nullptr, hipEvent_t stopped = nullptr) {
    const uint32_t reach = shape.reach;
    const size_t lds = shape.lds;
    // 6 waves per SIMD (<= 80 VGPRs; six workgroups of four waves per CU, 129 of its 160 KB of LDS): until round 5 that build spilled nine
    // registers across the walk and was slower than five waves at 88; with what the chain behind the walk needs worked out behind the walk
    // it fits in 79 and is level alone, 3-4 % faster with three frames in flight (profiles/round5_ab_six_waves.txt); 7 waves (72) spill sixteen
    if (stamps)
        hipLaunchKernelGGL((k_update_tiled<true, 6>), dim3(shape.groups_stamped), dim3(kTileThreads), lds, stream, points0, points1, normals, items, h, pos, res, recip, trunc, m, reach, adaptive ? 1u : 0u, items_capacity, stamps, pack_counter, pack_seen_host);
    else if (started && stopped)  // the events take the dispatch's own begin and end time stamps: the kernel's execution, without marker packets around it
        hipExtLaunchKernelGGL((k_update_tiled<false, 6>), dim3(shape.groups), dim3(kTileThreads), uint32_t(lds), stream, started, stopped, 0u, points0, points1, normals, items, h, pos,
                              res, recip, trunc, m, reach, adaptive ? 1u : 0u, items_capacity, stamps, pack_counter, pack_seen_host);
    else
        hipLaunchKernelGGL((k_update_tiled<false, 6>), dim3(shape.groups), dim3(kTileThreads), lds, stream, points0, points1, normals, items, h, pos, res, recip, trunc, m, reach, adaptive ? 1u : 0u, items_capacity, stamps, pack_counter, pack_seen_host);
}

}  // namespace chadhip

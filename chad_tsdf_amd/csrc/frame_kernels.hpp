// frame_kernels.hpp -- per-frame stages ahead of the TSDF update: voxel bounds, sort keys, gather + block
// segments, and normal estimation (reference: morton.hpp:59-102, normals.hpp:10-148).
//
// Processing order.  The reference sorts the whole frame by descending 63-bit Morton code.  Everything that
// depends on that order is local to a 4x4x4-voxel block (the greedy neighbourhood walk of normals.hpp:94-108
// never crosses a change of code >> 6, and every block's first neighbourhood starts at the block's first
// point), except one thing: the globally last point is never absorbed into a neighbourhood (normals.hpp:100).
// So the device sorts by a SHORT key -- the position of the voxel's block inside the frame's bounding box (whose
// corner is aligned to 8 voxels, so the block partition is that of the global code) and nothing else -- and finds the
// reference's "last point" (smallest global code, largest input index among equals) with a reduction.  A room-sized
// frame needs 2 radix passes instead of 8 (16 key bits; with the voxel's six code bits in the key it was 3).
// Inside a block the order must be the reference's -- descending voxel code, ties by ascending input index -- because
// the neighbourhoods are runs of that order: the radix sort is stable, so a block's points arrive in input order, and
// the normals kernel, which stages a block anyway, finishes the job with a stable counting sort by voxel (64 bins, in
// LDS) and writes the block out in that order for the update kernel (estimate_block).
// The ORDER OF THE BLOCKS is free, and is chosen for the update kernel: the block position is split into the
// position of its 2x2x2-block "super block" (8x8x8 voxels, the unit the update kernel stages in LDS) and three bits
// for the block inside it, so that the points of a super block are contiguous as well.
#pragma once
#include <type_traits>

#include "device_math.hpp"
#include "radix_sort.hpp"

namespace chadhip {

constexpr int kFrameBlocks = 512;     // fixed grid of the bounds / keys / gather kernels (2 workgroups per CU)
constexpr int kGatherChunk = 2048;    // points per gather step of one workgroup (1024: more blocks cross a chunk end and probe for their length; slower)
constexpr uint32_t kItemPoints = 1024; // an update work item never holds more points than this
constexpr uint32_t kBigItem = 384;     // items longer than this are handed out first (longest-first keeps the update kernel's tail short)
constexpr uint32_t kWorkGroups = 256; // update waves pull items from one of this many counters (blockIdx % kWorkGroups)
constexpr uint32_t kListStripes = 16; // stripes of the frame's block / work-item lists (FrameHeader::list_count)
constexpr uint32_t kSparsePointsPerBlock = 6;  // frames with fewer points per block than this take the scatter form of the update
constexpr uint32_t kFirstItems = 2048; // workgroups of the update kernel whose first work item k_frame_normals looks up for them (FrameHeader::first_item)
constexpr uint32_t kBigBlock = 256;   // point blocks with more points than this get a workgroup of four waves in k_frame_normals (a wave works a
                                      // block off in a chain proportional to its size: with one wave each, the few blocks next to the sensor -- a
                                      // thousand points and more -- set the kernel's duration: 45 us, 25 us without them)

// Per-frame scalars living in device memory (one per map; rewritten every insert).
struct FrameHeader {
    int vmin[3], vmax[3];       // voxel bounds of the usable points
    uint32_t n_usable;          // points kept (D2 of the oracle header)
    uint32_t axis_bits;         // bits of a usable sort key
    uint32_t sort_passes;       // radix passes needed: ceil((axis_bits + 1 if any point is unusable) / 8)
    uint32_t reserved1;
    unsigned long long mc_min;  // smallest global Morton code of the frame: its block holds the reference's last point (normals.hpp:100),
    uint32_t last_idx;          // (unused: that point is simply the last one of its block in the reference's order)
    uint32_t reserved0;
    uint32_t reserved2, reserved3;
    unsigned long long n_updates[kUpdateStripes * 16];  // voxel updates applied by the last insert, striped over cache lines
    uint32_t work_cursor[kWorkGroups * 16];  // per-group work counters of the update kernel, one per 64-byte line
    // Entries of the frame's three lists -- point blocks, update work items longer than kBigItem points, the shorter items --
    // per stripe, one counter per 64-byte line.  k_frame_gather appends a chunk's entries to the stripe of its sixteenth of the frame
    // with one returning atomic per list: on ONE counter per list the 489 workgroups of a 1 M-point frame queued for
    // ~25 ns each, half of that kernel's 26 us.  Stripe s of a list owns entries [s * stride, (s + 1) * stride) of its array,
    // stride = capacity / kListStripes + kGatherChunk; long items fill a stripe's range from the front, short ones from the back.
    uint32_t list_count[4][kListStripes * 16];
    // For the update kernel, written by the last workgroup of k_frame_normals (the lists are complete by then): where each stripe's
    // long / short work items begin when counted through the stripes in order ([list][kListStripes] = the list's total), whether the
    // frame is sparse, and the record of work item v for the first kFirstItems items -- workgroup b of the update kernel starts on
    // item b.  Looked up by every workgroup for itself these were three dependent round trips (stripe counts -> item record ->
    // points) in front of the first ray, 4.9 us of the kernel's 31 (in-kernel stamps); now it is two.
    uint32_t item_first[2][kListStripes + 1];
    uint32_t frame_sparse, reserved4;
    uint4 first_item[kFirstItems];
};
static_assert(offsetof(FrameHeader, first_item) % 16 == 0, "first_item[] is read with 16-byte loads");
// (the large point blocks -- more than kBigBlock points -- are listed a second time, from the back of the block list's stripes:
// k_frame_normals gives each of them a whole workgroup)
enum { kListBlocks = 0, kListLongItems = 1, kListShortItems = 2, kListBigBlocks = 3 };

__device__ __forceinline__ uint32_t list_total(const FrameHeader* h, int list) {
    uint32_t n = 0;
    for (uint32_t s = 0; s < kListStripes; s++) n += h->list_count[list][s * 16];
    return n;
}
// Entry i of a list, counted through the stripes in order.  The stripe counts are read once (sixteen independent loads,
// wave-uniform) and the search runs on registers: looked up straight from the header, every entry cost up to sixteen
// DEPENDENT loads (the update kernel went from 30 to 36 us that way).
struct ListIndex {
    uint32_t begin[kListStripes + 1];   // entries in the stripes before stripe s
    __device__ __forceinline__ void load(const FrameHeader* h, int list) {
        uint32_t count[kListStripes];
#pragma unroll
        for (uint32_t s = 0; s < kListStripes; s++) count[s] = h->list_count[list][s * 16];
        begin[0] = 0;
#pragma unroll
        for (uint32_t s = 0; s < kListStripes; s++) begin[s + 1] = begin[s] + count[s];
    }
    __device__ __forceinline__ uint32_t total() const { return begin[kListStripes]; }
    // stripe of entry i (return value) and its index inside the stripe
    __device__ __forceinline__ uint32_t locate(uint32_t i, uint32_t* index) const {
        uint32_t stripe = 0, first = 0;
#pragma unroll
        for (uint32_t s = 1; s < kListStripes; s++) { const bool later = i >= begin[s]; stripe = later ? s : stripe; first = later ? begin[s] : first; }
        *index = i - first;
        return stripe;
    }
};
__device__ __forceinline__ bool frame_is_sparse(const FrameHeader* h) { return h->n_usable < kSparsePointsPerBlock * list_total(h, kListBlocks); }

// Sparse frames (few points per 4x4x4 block, e.g. 2 cm voxels outdoors) take the scatter form of the update: with about
// one point per block, staging a voxel tile in LDS per block is pure overhead.  Decided on the device, per frame.

// Partial results of one workgroup, combined by the NEXT kernel's prologue (no same-address atomics: on gfx950
// those serialise at ~12 ns each, which for thousands of waves costs more than the whole pass).
struct FramePartial {
    int lo[3], hi[3];
    uint32_t kept, pad;
    unsigned long long mc_min;
};

// Per-frame reset of the header.  Runs as its own kernel only for
// empty frames; otherwise workgroup 0 of k_frame_keys does it (nothing before k_frame_gather reads the header).
__device__ __forceinline__ void frame_header_reset(FrameHeader* h) {
    for (uint32_t k = threadIdx.x; k < kUpdateStripes * 16; k += blockDim.x) h->n_updates[k] = 0;
    for (uint32_t k = threadIdx.x; k < kWorkGroups * 16; k += blockDim.x) h->work_cursor[k] = 0;
    for (uint32_t k = threadIdx.x; k < 4 * kListStripes * 16; k += blockDim.x) (&h->list_count[0][0])[k] = 0;
    if (threadIdx.x == 0) {
        h->vmin[0] = h->vmin[1] = h->vmin[2] = 0x7fffffff;
        h->vmax[0] = h->vmax[1] = h->vmax[2] = int(0x80000000);
        h->n_usable = 0; h->axis_bits = 0; h->sort_passes = 0; h->reserved1 = 0;
        h->mc_min = ~0ull; h->last_idx = 0; h->reserved0 = 0; h->reserved2 = 0; h->reserved3 = 0;
    }
}
__global__ void k_frame_init(FrameHeader* h) { frame_header_reset(h); }

// Reductions over the 64 lanes of a wave, the result in every lane: the scan network of wave_inclusive_sum (radix_sort.hpp) -- four shifts
// inside the 16-lane rows, row_bcast:15, row_bcast:31: lane 63 holds the result -- on the VALU, then one v_readlane.  (As xor butterflies
// over __shfl they were six ds_bpermute round trips each: forty-two in a row at the end of k_frame_bounds and again at the start of
// k_frame_keys, two microseconds of each kernel.)  A lane shifted in from outside its row, or a row the broadcast does not reach, keeps
// its own value: every step combines a value with itself or with another lane's.  All 64 lanes must execute them.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_keep(int v) { return __builtin_amdgcn_update_dpp(v, v, kCtrl, kRowMask, 0xf, false); }
template <class Op>
__device__ __forceinline__ int wave_reduce(int v, Op op) {
    v = op(v, dpp_keep<0x111, 0xf>(v));   // row_shr:1
    v = op(v, dpp_keep<0x112, 0xf>(v));   // row_shr:2
    v = op(v, dpp_keep<0x114, 0xf>(v));   // row_shr:4
    v = op(v, dpp_keep<0x118, 0xf>(v));   // row_shr:8
    v = op(v, dpp_keep<0x142, 0xa>(v));   // row_bcast:15 -> rows 1 and 3
    v = op(v, dpp_keep<0x143, 0xc>(v));   // row_bcast:31 -> rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min(int v) { return wave_reduce(v, [](int a, int b) { return min(a, b); }); }
__device__ __forceinline__ int wave_max(int v) { return wave_reduce(v, [](int a, int b) { return max(a, b); }); }
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return uint32_t(__builtin_amdgcn_readlane(int(wave_inclusive_sum(v)), 63)); }
template <int kCtrl, int kRowMask>
__device__ __forceinline__ unsigned long long dpp_keep_u64(unsigned long long v) {
    const uint32_t lo = uint32_t(dpp_keep<kCtrl, kRowMask>(int(uint32_t(v)))), hi = uint32_t(dpp_keep<kCtrl, kRowMask>(int(uint32_t(v >> 32))));
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    unsigned long long t;
    t = dpp_keep_u64<0x111, 0xf>(v); v = t < v ? t : v;
    t = dpp_keep_u64<0x112, 0xf>(v); v = t < v ? t : v;
    t = dpp_keep_u64<0x114, 0xf>(v); v = t < v ? t : v;
    t = dpp_keep_u64<0x118, 0xf>(v); v = t < v ? t : v;
    t = dpp_keep_u64<0x142, 0xa>(v); v = t < v ? t : v;
    t = dpp_keep_u64<0x143, 0xc>(v); v = t < v ? t : v;
    return (unsigned long long)uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v)), 63)) | ((unsigned long long)uint32_t(__builtin_amdgcn_readlane(int(uint32_t(v >> 32)), 63)) << 32);
}

// min over the lanes l .. 63, in every lane l: four shifts inside the 16-lane rows (a lane shifted in from outside its row keeps its
// value), then the minima of the rows behind -- they stand in the rows' first lanes -- through scalar registers.  All 64 lanes.
__device__ __forceinline__ uint32_t wave_suffix_min(uint32_t v) {
    v = min(v, uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x101, 0xf, 0xf, false)));   // row_shl:1
    v = min(v, uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x102, 0xf, 0xf, false)));   // row_shl:2
    v = min(v, uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x104, 0xf, 0xf, false)));   // row_shl:4
    v = min(v, uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), 0x108, 0xf, 0xf, false)));   // row_shl:8
    const uint32_t r3 = uint32_t(__builtin_amdgcn_readlane(int(v), 48)), r2 = min(uint32_t(__builtin_amdgcn_readlane(int(v), 32)), r3),
                   r1 = min(uint32_t(__builtin_amdgcn_readlane(int(v), 16)), r2);
    const uint32_t row = (threadIdx.x & 63u) >> 4;
    return min(v, row == 0u ? r1 : row == 1u ? r2 : row == 2u ? r3 : 0xffffffffu);
}

__device__ __forceinline__ F3 load_point(const float* __restrict__ xyz, size_t i) { return {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}; }

// The frame's points four at a time: three 16-byte loads per lane (48 contiguous bytes) instead of twelve 4-byte ones at a stride
// of 12 -- when the array is 16-byte aligned (any hipMalloc-ed buffer).  visit(i, p) is called for every point; the last n % 4
// points, or all of them when the array is not aligned, go one by one.
template <class Visit>
__device__ __forceinline__ void for_each_point(const float* __restrict__ xyz, uint32_t n, Visit visit) {
    const size_t t = size_t(blockIdx.x) * blockDim.x + threadIdx.x, threads = size_t(gridDim.x) * blockDim.x;
    const uint32_t quads = (reinterpret_cast<uintptr_t>(xyz) & 15u) == 0u ? n / 4u : 0u;
    const float4* __restrict__ wide = reinterpret_cast<const float4*>(xyz);
    for (size_t g = t; g < quads; g += threads) {
        const float4 a = wide[3 * g], b = wide[3 * g + 1], c = wide[3 * g + 2];
        visit(4 * g, F3{a.x, a.y, a.z}); visit(4 * g + 1, F3{a.w, b.x, b.y}); visit(4 * g + 2, F3{b.z, b.w, c.x}); visit(4 * g + 3, F3{c.y, c.z, c.w});
    }
    for (size_t i = size_t(quads) * 4 + t; i < n; i += threads) visit(i, load_point(xyz, i));
}

// Stage 1: voxel bounds and the count of usable points, one partial per workgroup.
__global__ void __launch_bounds__(256)
k_frame_bounds(const float* __restrict__ xyz, uint32_t n, F3 pos, float recip, FramePartial* __restrict__ part, uint32_t* __restrict__ digit_total) {
    __shared__ int red[4][7];
    // the sort's digit totals, which the keys kernel adds into (saves the sort a kernel of its own for them)
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < kBaseStripes * kMaxPasses * kRadixMax; k += gridDim.x * blockDim.x) digit_total[k] = 0u;
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {int(0x80000000), int(0x80000000), int(0x80000000)};
    uint32_t kept = 0;
    for_each_point(xyz, n, [&](size_t, F3 p) {
        if (point_usable(p, pos, recip)) {
            const I3 v = voxel_of(p, recip);
            lo[0] = min(lo[0], v.x); lo[1] = min(lo[1], v.y); lo[2] = min(lo[2], v.z);
            hi[0] = max(hi[0], v.x); hi[1] = max(hi[1], v.y); hi[2] = max(hi[2], v.z);
            kept++;
        }
    });
#pragma unroll
    for (int a = 0; a < 3; a++) { lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]); }
    kept = wave_sum(kept);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { red[wave][a] = lo[a]; red[wave][3 + a] = hi[a]; }
        red[wave][6] = int(kept);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        FramePartial p;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            p.lo[a] = min(min(red[0][a], red[1][a]), min(red[2][a], red[3][a]));
            p.hi[a] = max(max(red[0][3 + a], red[1][3 + a]), max(red[2][3 + a], red[3][3 + a]));
        }
        p.kept = uint32_t(red[0][6]) + uint32_t(red[1][6]) + uint32_t(red[2][6]) + uint32_t(red[3][6]);
        p.pad = 0; p.mc_min = ~0ull;
        part[blockIdx.x].lo[0] = p.lo[0]; part[blockIdx.x].lo[1] = p.lo[1]; part[blockIdx.x].lo[2] = p.lo[2];
        part[blockIdx.x].hi[0] = p.hi[0]; part[blockIdx.x].hi[1] = p.hi[1]; part[blockIdx.x].hi[2] = p.hi[2];
        part[blockIdx.x].kept = p.kept;
    }
}

struct LocalFrame {  // geometry of the frame-local sort key
    int ox, oy, oz;          // super-block-aligned (8 voxels) corner of the frame's voxel bounds
    uint32_t nx, ny, nz;     // super blocks along each axis inside the bounds
    uint32_t bits, kept;     // bits of a usable key; usable points
    int lo[3], hi[3];
};

// Combine the kFrameBlocks partial bounds (every workgroup of the keys kernel does this redundantly: 512 x 28 B
// from L2) and derive the frame-local code geometry.
__device__ __forceinline__ LocalFrame combine_bounds(const FramePartial* __restrict__ part) {
    __shared__ int red[4][7];
    __shared__ LocalFrame shared_frame;
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {int(0x80000000), int(0x80000000), int(0x80000000)};
    uint32_t kept = 0;
    for (uint32_t k = threadIdx.x; k < kFrameBlocks; k += blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; a++) { lo[a] = min(lo[a], part[k].lo[a]); hi[a] = max(hi[a], part[k].hi[a]); }
        kept += part[k].kept;
    }
#pragma unroll
    for (int a = 0; a < 3; a++) { lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]); }
    kept = wave_sum(kept);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { red[wave][a] = lo[a]; red[wave][3 + a] = hi[a]; }
        red[wave][6] = int(kept);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        LocalFrame f;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            f.lo[a] = min(min(red[0][a], red[1][a]), min(red[2][a], red[3][a]));
            f.hi[a] = max(max(red[0][3 + a], red[1][3 + a]), max(red[2][3 + a], red[3][3 + a]));
        }
        f.kept = uint32_t(red[0][6]) + uint32_t(red[1][6]) + uint32_t(red[2][6]) + uint32_t(red[3][6]);
        f.ox = f.lo[0] & ~7; f.oy = f.lo[1] & ~7; f.oz = f.lo[2] & ~7;  // align down to a super-block corner
        // super blocks 0 .. (hi - o) >> 3 along each axis, at most 2^18 of them (beyond 2^21 voxels the reference's code aliases as
        // well, morton.hpp:22); their index is x + nx * (y + ny * z): mixed radix, not bit fields -- a 25 x 20 x 8 room is
        // 4000 < 2^12 super blocks where bit fields need 13 bits, and that bit decides between two sort passes and three
        auto extent = [](int span_voxels) { return min(uint32_t(span_voxels) >> 3, (1u << 18) - 1u) + 1u; };
        f.nx = f.kept ? extent(f.hi[0] - f.ox) : 1u; f.ny = f.kept ? extent(f.hi[1] - f.oy) : 1u; f.nz = f.kept ? extent(f.hi[2] - f.oz) : 1u;
        const unsigned long long last = (unsigned long long)f.nx * f.ny * f.nz - 1ull;
        f.bits = (last ? 64u - uint32_t(__clzll((long long)last)) : 0u) + 3u;
        shared_frame = f;
    }
    __syncthreads();
    return shared_frame;
}

// Stage 2: sort keys.  key = [super-block index][block in super block : 3 bits], the super-block index being its
// x + nx * (y + ny * z) position inside the frame's bounds.  Ascending keys therefore group the points by super block,
// then by block; inside a block they keep input order (the sort is stable) and the normals kernel orders them by voxel.
// The order of the blocks themselves is irrelevant (see the header of this file), so it need not be Morton order -- and
// a room-sized frame needs 15 key bits + 1 (2 radix passes) instead of 3*8 + 1.  Unusable points get 2^bits, above every
// usable key, so they end up behind the n_usable usable points.
__global__ void __launch_bounds__(256)
k_frame_keys(const float* __restrict__ xyz, uint32_t n, F3 pos, float recip, FrameHeader* __restrict__ h,
             FramePartial* __restrict__ part, uint64_t* __restrict__ keys,
             uint32_t* __restrict__ digit_total, uint32_t* __restrict__ tile_status, uint32_t status_words,
             const uint32_t* __restrict__ live_blocks, uint32_t* __restrict__ live_blocks_host, uint32_t digit_bits /* 8 or 9: what the sort passes were launched with */) {
    if (blockIdx.x == 0) frame_header_reset(h);
    if (blockIdx.x == 0) {  // host-mapped: capacity growth hint (the live-brick counter is striped, one stripe per thread)
        static_assert(kCountStripes == 256, "one stripe per thread of this workgroup");
        __shared__ uint32_t live_sum[4];
        const uint32_t live = wave_sum(live_blocks[threadIdx.x * 16]);
        if ((threadIdx.x & 63) == 0) live_sum[threadIdx.x >> 6] = live;
        __syncthreads();
        if (threadIdx.x == 0) *live_blocks_host = live_sum[0] + live_sum[1] + live_sum[2] + live_sum[3];
    }
    const LocalFrame f = combine_bounds(part);
    __shared__ unsigned long long wave_mc[4];
    // digit histograms of all radix passes, taken while the key is in a register (radix_sort.hpp)
    __shared__ uint32_t hist[kMaxPasses][kRadixMax];
    const uint32_t key_bits = f.bits + (f.kept < n ? 1u : 0u), radix = 1u << digit_bits;
    const uint32_t passes = (key_bits + digit_bits - 1) / digit_bits;
    // clear the radix sort's look-back table for this frame (saves a memset launch): the words of the passes that will run
    // (layout [pass][tile][digit], radix_sort.hpp) and the ticket words behind the table
    {
        const size_t per_pass = (status_words - 64) / kMaxPasses, used = per_pass * passes;
        for (size_t k = size_t(blockIdx.x) * blockDim.x + threadIdx.x; k < used; k += size_t(gridDim.x) * blockDim.x) tile_status[k] = 0u;
        if (blockIdx.x == 0 && threadIdx.x < 64) tile_status[status_words - 64 + threadIdx.x] = 0u;
    }
    for (uint32_t k = threadIdx.x; k < kMaxPasses * kRadixMax; k += blockDim.x) (&hist[0][0])[k] = 0;
    __syncthreads();
    const uint64_t top = 1ull << f.bits;
    // The frame's smallest Morton code (its block holds the reference's last point): the running minimum is kept as a VOXEL and
    // compared in Morton order without encoding -- the axis whose coordinates differ in the highest bit decides, z before y before x
    // among equals (z is the most significant bit of every triple) -- and encoded once per thread at the end: three 21-bit spreads
    // per point were 95 of this kernel's 155 instructions per 64 points.
    bool any_usable = false;
    uint32_t min_x = 0, min_y = 0, min_z = 0;   // biased as morton_encode biases them: 2^20 + v, 21 bits
    auto msb_below = [](uint32_t a, uint32_t b) { return a < b && a < (a ^ b); };   // the highest set bit of a lies below that of b
    for_each_point(xyz, n, [&](size_t i, F3 p) {
        uint64_t key = top;
        if (point_usable(p, pos, recip)) {
            const I3 v = voxel_of(p, recip);
            const uint32_t bx = ((1u << 20) + uint32_t(v.x)) & 0x1fffffu, by = ((1u << 20) + uint32_t(v.y)) & 0x1fffffu, bz = ((1u << 20) + uint32_t(v.z)) & 0x1fffffu;
            const uint32_t dz = bz ^ min_z, dy = by ^ min_y, dx = bx ^ min_x;
            uint32_t deciding = dz;
            bool below = bz < min_z;
            if (msb_below(deciding, dy)) { deciding = dy; below = by < min_y; }
            if (msb_below(deciding, dx)) { deciding = dx; below = bx < min_x; }
            if (!any_usable || below) { min_x = bx; min_y = by; min_z = bz; }
            any_usable = true;
            const uint32_t rx = uint32_t(v.x - f.ox), ry = uint32_t(v.y - f.oy), rz = uint32_t(v.z - f.oz);  // voxel inside the frame's bounds
            const uint64_t sx = min(rx >> 3, f.nx - 1u), sy = min(ry >> 3, f.ny - 1u), sz = min(rz >> 3, f.nz - 1u);   // (clamped only beyond 2^21 voxels)
            const uint32_t inner = ((rx >> 2) & 1u) | (((ry >> 2) & 1u) << 1) | (((rz >> 2) & 1u) << 2);
            key = ((sx + f.nx * (sy + f.ny * sz)) << 3) | inner;
        }
        keys[i] = key;
        for (uint32_t pass = 0; pass < passes; pass++) atomicAdd(&hist[pass][uint32_t(key >> (pass * digit_bits)) & (radix - 1u)], 1u);
    });
    __syncthreads();
    // digit totals of the radix passes (RadixSortBuffers::digit_base): one of kBaseStripes copies per workgroup, zero counts skipped
    for (uint32_t p = 0; p < passes; p++)
        for (uint32_t d = threadIdx.x; d < radix; d += blockDim.x) {
            const uint32_t c = hist[p][d];
            if (c) atomicAdd(&digit_total[((blockIdx.x & (kBaseStripes - 1u)) * kMaxPasses + p) * radix + d], c);
        }
    unsigned long long mc_min = any_usable ? (unsigned long long)(spread21(min_x) | (spread21(min_y) << 1) | (spread21(min_z) << 2)) : ~0ull;   // morton_encode of the minimum
    mc_min = wave_min_u64(mc_min);
    if ((threadIdx.x & 63) == 0) wave_mc[threadIdx.x >> 6] = mc_min;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = wave_mc[0];
        for (int w = 1; w < 4; w++) m = wave_mc[w] < m ? wave_mc[w] : m;
        part[blockIdx.x].mc_min = m;
        if (blockIdx.x == 0) {
            for (int a = 0; a < 3; a++) { h->vmin[a] = f.lo[a]; h->vmax[a] = f.hi[a]; }
            h->n_usable = f.kept;
            h->axis_bits = f.bits;
            h->sort_passes = passes;  // the extra top bit is only needed when some point is unusable
            live_blocks_host[1] = passes;    // host-mapped hints for the next frames: how many single-pass sort kernels this frame needed ...
            live_blocks_host[2] = key_bits;  // ... and how wide its key was (the host picks the digit width and the launch count from it)
        }
    }
}

// Length of the run of points of `block` (= the sort key of a usable point) starting at `first`, capped at `limit`.
// Two dependent loads instead of one per 64 points: lanes first probe the starts of the next 64 chunks of 64
// points, then the points of the chunk where the run ends.  Wave-uniform result.
__device__ __forceinline__ uint32_t run_length(const uint64_t* __restrict__ keys, uint32_t first, uint32_t limit, uint64_t block, uint32_t lane) {
    uint32_t len = 0;
    for (;;) {
        const uint32_t probe = first + len + 64u * (lane + 1u);                 // first point of chunk lane+1
        const uint64_t more = __ballot(probe < limit && keys[probe] == block);
        const uint32_t full = more == ~0ull ? 64u : uint32_t(__ffsll((unsigned long long)~more)) - 1u;  // chunks 1..full start inside the run
        len += 64u * full;
        if (full == 64u) continue;                                               // the run is longer than 4160 points: go on
        const uint32_t i = first + len + lane;                                   // the chunk where the run ends
        const uint64_t same = __ballot(i < limit && keys[i] == block);
        return len + (same == ~0ull ? 64u : uint32_t(__ffsll((unsigned long long)~same)) - 1u);
    }
}

// Stage 3 (after the sort, which has carried the points along as float4 (x, y, z, input index)): list (a) the start of every point block (run of equal code >> 6) in seg_start[] for the
// normal estimation and (b) the update work items (start, length, super-block code) in items[]: the super blocks (runs
// of equal code >> 9) cut again at every kItemPoints-th point, so that no item is longer than kItemPoints.  A workgroup collects its entries in
// LDS (block heads as a list, item starts as a 2048-bit mask whose next set bit gives an item's length) and
// reserves output space with ONE atomic per list and chunk; the lists are in no particular order.
// The sorted (key, idx) pairs are in sort buffer (sort_passes & 1).
// Its prologue is the sort's catch-all (radix_sort.hpp, radix_rest_points: a frame whose key has outgrown the launches that were
// enqueued for it -- normally one load and a comparison); kBits: the digit width of this frame's sort.
template <int kBits>
__global__ void __launch_bounds__(256, 4)   // (four waves per SIMD, as before the sort's catch-all moved in: <= 128 VGPRs)
k_frame_gather(float recip, FrameHeader* __restrict__ h, const FramePartial* __restrict__ part,
               const uint64_t* keys0, const uint64_t* keys1, const float4* points0, const float4* points1,
               uint2* __restrict__ segments, uint4* __restrict__ items, uint32_t list_stride,
               uint32_t* __restrict__ consumed, uint32_t consumed_seq, uint32_t big_block, RadixRest rest) {
    constexpr uint32_t kWords = kGatherChunk / 32;
    if (rest.participants) {
        const uint32_t limit = min(uint32_t(kMaxPasses), h->sort_passes);
        if (rest.first_pass < limit) radix_rest_points<kBits>(rest, limit);   // (the same decision in every workgroup)
    }
    // the sort is over (stream order), so the caller's raw points -- read by its first pass only -- are no longer needed: tell the
    // host (host-mapped word; chad_hip.hip, prepare_slot).  The sort's later launches store the same number earlier when there are any.
    if (consumed && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(consumed, consumed_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __shared__ uint32_t mark_bits[kWords], head_bits[kWords];  // bit k: point k of the chunk starts a work item / a point block
    __shared__ uint32_t word_rank[2][kWords], word_next[2][kWords];
    __shared__ uint32_t base_heads, n_big, n_small, base_big, base_small, trail_extra;
    __shared__ unsigned long long wave_mc[4], block_mc_min;
    const uint32_t n = h->n_usable;            // (loaded beside the partials below: one round trip, not two)
    const bool odd = h->sort_passes & 1u;
    // frame-wide smallest code from the keys kernel's partials
    {
        unsigned long long m = ~0ull;
        for (uint32_t k = threadIdx.x; k < kFrameBlocks; k += blockDim.x) { const unsigned long long t = part[k].mc_min; m = t < m ? t : m; }
        m = wave_min_u64(m);
        if ((threadIdx.x & 63) == 0) wave_mc[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = wave_mc[0];
            for (int w = 1; w < 4; w++) m = wave_mc[w] < m ? wave_mc[w] : m;
            block_mc_min = m;
            if (blockIdx.x == 0) h->mc_min = m;
        }
        __syncthreads();
    }
    const unsigned long long mc_min = block_mc_min;
    const uint64_t* keys = odd ? keys1 : keys0;
    const float4* pts = odd ? points1 : points0;   // the sort carries the points along: its last pass left them in processing order
    const uint32_t chunks = (n + kGatherChunk - 1) / kGatherChunk;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t stripe = c / ((chunks + kListStripes - 1) / kListStripes);   // consecutive chunks share a stripe: the lists keep the frame's order
        if (threadIdx.x < kWords) { mark_bits[threadIdx.x] = 0u; head_bits[threadIdx.x] = 0u; }
        if (threadIdx.x == 0) { n_big = 0u; n_small = 0u; }
        __syncthreads();
        constexpr uint32_t kPer = kGatherChunk / 256;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint32_t i = c * kGatherChunk + j * 256 + threadIdx.x;
            if (i < n) {
                const uint64_t kcur = keys[i], kprev = i ? keys[i - 1] : ~0ull;   // nothing precedes point 0: it opens a block
                const bool head = kcur != kprev, super_head = (kcur >> 3) != (kprev >> 3);   // key = [super block][block in it : 3]
                if (head) atomicOr(&head_bits[(j * 256 + threadIdx.x) >> 5], 1u << (threadIdx.x & 31u));
                if (super_head || (i % kItemPoints) == 0) atomicOr(&mark_bits[(j * 256 + threadIdx.x) >> 5], 1u << (threadIdx.x & 31u));
            }
        }
        __syncthreads();
        // work items: every mark opens an item that runs to the next mark, or to the end of the chunk (chunks start at a
        // multiple of kItemPoints, i.e. at a mark, so items never span chunks)
        const uint32_t chunk_len = min(uint32_t(kGatherChunk), n - c * kGatherChunk);
        // rank of every mark among the chunk's marks and position of the next one, by word of the bit masks (kWords = 64 = one
        // word per lane): wave 0 for the items, wave 1 for the point blocks -- a prefix sum of the popcounts and a suffix
        // minimum of the words' first marks (two serial 64-step loops per list before)
        static_assert(kWords == 64, "one bit-mask word per lane");
        if (threadIdx.x < 128) {
            const uint32_t which = threadIdx.x >> 6, lane = threadIdx.x & 63u;
            const uint32_t word = (which ? head_bits : mark_bits)[lane];
            const uint32_t count = uint32_t(__popc(word));
            const uint32_t incl = wave_inclusive_sum(count);
            word_rank[which][lane] = incl - count;
            const uint32_t first_mark = wave_suffix_min(word ? lane * 32u + uint32_t(__ffs(int(word)) - 1) : chunk_len);   // chunk_len = "none"
            const uint32_t after = __shfl_down(first_mark, 1);   // first mark in a later word
            word_next[which][lane] = lane == 63u ? chunk_len : after;
            if (which && lane == 63u) base_heads = stripe * list_stride + (incl ? atomicAdd(&h->list_count[kListBlocks][stripe * 16u], incl) : 0u);
        } else if (threadIdx.x < 192) {
            // wave 2 meanwhile: the chunk's last point block may go on in the next chunks -- how far?  (two dependent loads; the
            // normals kernel used to find out itself, one wave per such block, and could not tell the large blocks apart beforehand)
            const uint32_t lane = threadIdx.x & 63u, word = head_bits[lane];
            const int last_in_word = word ? int(lane * 32u) + 31 - __clz(int(word)) : -1;
            const int last_head = wave_max(last_in_word);
            uint32_t extra = 0;
            if (last_head >= 0 && c + 1 < chunks) extra = run_length(keys, (c + 1u) * kGatherChunk, n, keys[c * kGatherChunk + uint32_t(last_head)], lane);
            if (lane == 0) trail_extra = extra;
        }
        __syncthreads();
        // the items of this chunk, long and short ones counted apart (order inside a class does not matter) ...
        uint32_t item_len[kGatherChunk / 256], item_slot[kGatherChunk / 256];
#pragma unroll
        for (uint32_t j = 0; j < kGatherChunk / 256; j++) {
            const uint32_t k = j * 256 + threadIdx.x, bit = k & 31u, word = mark_bits[k >> 5];
            item_len[j] = 0u; item_slot[j] = 0u;
            if (k < chunk_len && ((word >> bit) & 1u)) {
                const uint32_t higher = bit == 31u ? 0u : word >> (bit + 1u);
                const uint32_t next = higher ? k + uint32_t(__ffs(int(higher))) : word_next[0][k >> 5];
                item_len[j] = min(next, chunk_len) - k;
                item_slot[j] = atomicAdd(item_len[j] > kBigItem ? &n_big : &n_small, 1u);
            }
        }
        __syncthreads();
        // ... each class reserving its part of the list with one atomic per chunk: long items from the front, short ones from the back
        // (by lanes of two different waves: from one thread the second atomic was issued when the first had returned -- two round trips in a row)
        if (threadIdx.x == 0) base_big = stripe * list_stride + (n_big ? atomicAdd(&h->list_count[kListLongItems][stripe * 16u], n_big) : 0u);
        if (threadIdx.x == 64) base_small = n_small ? atomicAdd(&h->list_count[kListShortItems][stripe * 16u], n_small) : 0u;
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < kGatherChunk / 256; j++) {
            const uint32_t k = j * 256 + threadIdx.x, bit = k & 31u;
            if (k >= chunk_len) continue;
            const uint32_t word = mark_bits[k >> 5], hword = head_bits[k >> 5];
            if (!(((word | hword) >> bit) & 1u)) continue;
            const float4 q = pts[c * kGatherChunk + k];
            const I3 v = voxel_of(F3{q.x, q.y, q.z}, recip);
            const uint64_t code = morton_encode(v.x, v.y, v.z);   // only the few threads that open a block or an item get here
            if ((word >> bit) & 1u) {
                const uint64_t corner = code & ~511ull;  // corner of the 8x8x8 super block
                const uint32_t at = item_len[j] > kBigItem ? base_big + item_slot[j] : (stripe + 1u) * list_stride - 1u - (base_small + item_slot[j]);
                items[at] = make_uint4(c * kGatherChunk + k, item_len[j], uint32_t(corner), uint32_t(corner >> 32));
            }
            if ((hword >> bit) & 1u) {
                // a point block: (start, length | "holds the frame's last point" << 31).  The chunk's last block may go on in the
                // next chunks: wave 2 has measured how far
                const uint32_t rank = word_rank[1][k >> 5] + __popc(hword & ((1u << bit) - 1u));
                const uint32_t higher = bit == 31u ? 0u : hword >> (bit + 1u);
                const uint32_t next = higher ? k + uint32_t(__ffs(int(higher))) : word_next[1][k >> 5];
                const uint32_t len = next < chunk_len ? next - k : chunk_len - k + trail_extra;
                const uint2 record = make_uint2(c * kGatherChunk + k, len | ((code >> 6) == (mc_min >> 6) ? 0x80000000u : 0u));
                segments[base_heads + rank] = record;
                if (len > big_block)   // listed again among the large blocks, from the back of the stripe (a hundred or so per frame)
                    segments[(stripe + 1u) * list_stride - 1u - atomicAdd(&h->list_count[kListBigBlocks][stripe * 16u], 1u)] = record;
            }
        }
        __syncthreads();
    }
}

// ---- Stage 4: normals (normals.hpp:10-148) -------------------------------------------------------------------
// One wave per point block (four for the large ones).  The block's <= 64 voxel runs are counted into LDS, the greedy neighbourhood
// walk runs over those 64 counters (not over the points), then the neighbourhoods are fitted with the reference's two-pass f64
// "plane from points" -- teams of 8 or 16 lanes per neighbourhood, the whole wave for the largest -- and every point picks up its
// neighbourhood's normal (or the per-point fallback when the neighbourhood has < 8 points).

constexpr uint32_t kMinPoints = 8;    // normals.hpp:88
constexpr int kMaxNormalWaves = 4;    // waves that may share a point block (k_frame_normals' workgroup)
constexpr uint32_t kStagePoints = 256;  // points of a block kept in LDS for the fits (4 KiB)
constexpr uint32_t kTeamPoints = 32;    // neighbourhoods of 32..255 points are fitted by a 16-lane team (4 at a time), those of 8..31 points by an 8-lane team (8 at a time)
constexpr uint32_t kCoopPoints = 256;   // neighbourhoods at least this large are fitted by the whole wave

constexpr uint32_t kCovSlots = 21;       // neighbourhoods whose covariance terms fit into the counting sort's 1 KiB (a wave on its own; a team of waves uses a neighbour's scratch)
struct NormalScratch {         // per wave, in LDS
    uint32_t group_of[64];     // neighbourhood of voxel v
    uint32_t g_first[64];      // first point (offset in the segment) of neighbourhood g
    uint32_t g_size[64];
    float g_normal[64][4];     // fitted normal; [3] != 0 marks "fit" (else per-point fallback)
    uint32_t team_list[64];    // neighbourhoods fitted by teams: the 8-lane ones from the front, the 16-lane ones from the back
    union {
        uint32_t run[kMaxNormalWaves][64];   // counting sort by voxel: wave w's points per voxel, then the next free position of voxel v for wave w
        double cov[kCovSlots][6];            // afterwards: covariance terms of fitted neighbourhoods waiting for their plane (estimate_block)
    };
    uint32_t cov_group[kCovSlots + 3];       // whose they are
    float4 stage[kStagePoints];  // the block's first points IN VOXEL ORDER (x, y, z, voxel inside the block as an integer)
};

// normals.hpp:39-79 -- normal from the covariance terms: three determinant-weighted axis directions, each aligned
// with the running sum, then normalised in f64 and cast to f32.
__device__ __forceinline__ F3 plane_from_covariance(double xx, double xy, double xz, double yy, double yz, double zz) {
    double wx = 0, wy = 0, wz = 0;
    {
        const double det = yy * zz - yz * yz, ax = det, ay = xz * yz - xy * zz, az = xy * yz - xz * yy;
        double w = det * det;
        if (wx * ax + wy * ay + wz * az < 0.0) w = -w;
        wx += ax * w; wy += ay * w; wz += az * w;
    }
    {
        const double det = xx * zz - xz * xz, ax = xz * yz - xy * zz, ay = det, az = xy * xz - yz * xx;
        double w = det * det;
        if (wx * ax + wy * ay + wz * az < 0.0) w = -w;
        wx += ax * w; wy += ay * w; wz += az * w;
    }
    {
        const double det = xx * yy - xy * xy, ax = xy * yz - xz * yy, ay = xy * xz - yz * xx, az = det;
        double w = det * det;
        if (wx * ax + wy * ay + wz * az < 0.0) w = -w;
        wx += ax * w; wy += ay * w; wz += az * w;
    }
    const double inv = 1.0 / sqrt(wx * wx + wy * wy + wz * wz);  // :79
    return {float(wx * inv), float(wy * inv), float(wz * inv)};
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The whole wave fits one LARGE neighbourhood (a dense voxel near the sensor holds hundreds of points): lanes
// stride over the points with coalesced 16-byte loads and the nine f64 moments are combined by butterfly
// reductions.  The summation order differs from the reference's sequential loop, i.e. results agree to f64
// rounding (~1e-16 relative), far inside the 1e-4 bar.  Every lane returns the same normal.
template <class Ptr>
__device__ __forceinline__ F3 fit_plane_wave(Ptr pts, uint32_t count, uint32_t lane) {
    double cx = 0, cy = 0, cz = 0;
    for (uint32_t k = lane; k < count; k += 64) { const float4 q = pts[k]; cx += double(q.x); cy += double(q.y); cz += double(q.z); }
    const double recip = 1.0 / double(count);
    cx = wave_sum_f64(cx) * recip; cy = wave_sum_f64(cy) * recip; cz = wave_sum_f64(cz) * recip;
    double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    for (uint32_t k = lane; k < count; k += 64) {
        const float4 q = pts[k];
        const double rx = double(q.x) - cx, ry = double(q.y) - cy, rz = double(q.z) - cz;
        xx += rx * rx; xy += rx * ry; xz += rx * rz; yy += ry * ry; yz += ry * rz; zz += rz * rz;
    }
    xx = wave_sum_f64(xx); xy = wave_sum_f64(xy); xz = wave_sum_f64(xz);
    yy = wave_sum_f64(yy); yz = wave_sum_f64(yz); zz = wave_sum_f64(zz);
    return plane_from_covariance(xx * recip, xy * recip, xz * recip, yy * recip, yz * recip, zz * recip);
}

// Sum over the 16 lanes of a DPP row, result in every lane of the row: four rotate-and-add steps on the VALU
// (row_ror:8,4,2,1), no LDS round trip.  Must be executed by all 64 lanes.
template <int kCtrl>
__device__ __forceinline__ double dpp_move_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double team_sum_f64(double v) {
    v += dpp_move_f64<0x128>(v);  // row_ror:8
    v += dpp_move_f64<0x124>(v);  // row_ror:4
    v += dpp_move_f64<0x122>(v);  // row_ror:2
    v += dpp_move_f64<0x121>(v);  // row_ror:1
    return v;
}

// The same over the 8 lanes of a half row: mirror the half, then swap neighbours, then pairs.  Every lane of the eight ends up with
// the same bits (each step adds the same two partial sums in one or the other order).
__device__ __forceinline__ double team8_sum_f64(double v) {
    v += dpp_move_f64<0x141>(v);  // row_half_mirror
    v += dpp_move_f64<0xB1>(v);   // quad_perm:[1,0,3,2]
    v += dpp_move_f64<0x4E>(v);   // quad_perm:[2,3,0,1]
    return v;
}
template <uint32_t kWidth> __device__ __forceinline__ double team_sum(double v) { return kWidth == 8u ? team8_sum_f64(v) : team_sum_f64(v); }

constexpr int kNormalWaves = 4;            // waves per workgroup of k_frame_normals
constexpr uint32_t kNormalBigGroups = 512;  // its first workgroups (at most): one large point block each at a time, all four waves on it (scripts/sweep_big_blocks.sh)

__device__ __forceinline__ uint32_t read_lane(uint32_t v, uint32_t l) { return uint32_t(__builtin_amdgcn_readlane(int(v), int(l))); }

// LDS operations of a wave complete in issue order: between a wave's own writes and reads a compiler barrier and a drained
// LDS counter are enough (what __syncthreads() amounts to in a one-wave workgroup)
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Normals of one point block (normals.hpp:81-148 restricted to the block: its walk never leaves it).  kTeam false: one wave does
// everything (the other waves of the workgroup work on other blocks, nothing is shared).  kTeam true: the whole workgroup of four
// waves -- the two passes over the points, the team fits and the whole-wave fits are dealt over the waves; the greedy walk is
// repeated by every wave (it runs in registers on the 64 voxel counters) and only wave 0 publishes it.
// low six code bits of a point's voxel: its place inside the 4x4x4 block (x -> bit 0, y -> bit 1, z -> bit 2, twice)
__device__ __forceinline__ uint32_t voxel_in_block(float4 q, float recip) {
    const I3 v = voxel_of(F3{q.x, q.y, q.z}, recip);
    const uint32_t x = uint32_t(v.x) & 3u, y = uint32_t(v.y) & 3u, z = uint32_t(v.z) & 3u;
    return (x & 1u) | ((y & 1u) << 1) | ((z & 1u) << 2) | ((x >> 1) << 3) | ((y >> 1) << 4) | ((z >> 1) << 5);
}

template <bool kTeam>
__device__ __forceinline__ void estimate_block(NormalScratch& s, const uint2 record, const float4* __restrict__ pts, float4* __restrict__ ordered, float recip,
                                               F3 pos, float4* __restrict__ normals, uint32_t ablate) {
    const uint32_t lane = threadIdx.x & 63u, wave = kTeam ? threadIdx.x >> 6 : 0u;
    const uint32_t tid = kTeam ? threadIdx.x : lane, stride = kTeam ? 64u * kNormalWaves : 64u, waves = kTeam ? uint32_t(kNormalWaves) : 1u;
    auto sync = [&]() { if (kTeam) __syncthreads(); else wave_lds_fence(); };
    const uint32_t first = record.x, len = record.y & 0x7fffffffu;
    if ((ablate >> 16) && len > (ablate >> 16)) return;   // timing experiment: blocks above a size are skipped (results are wrong)
    // normals.hpp:100 -- the frame's last point never joins a neighbourhood.  It is the point with the smallest code and, among
    // those, the largest input index: the LAST point of this block in the reference's order.
    const bool holds_last = (record.y >> 31) != 0u;
    const uint32_t grouped = holds_last ? len - 1 : len;

    // ---- the block in the reference's order: a stable counting sort by voxel, descending (the radix sort delivered input order) ----
    // Every wave takes a contiguous share of the block (a multiple of 64 points): the shares keep their order, and inside a
    // share the 64-point steps do.  A share of at most kKeep steps (every block of up to 256 points, every team block of up to
    // 1024) is read ONCE: the points stay in registers and each gets its rank among the share's points of its voxel as the
    // steps go by (the lanes holding one voxel rank themselves by ballots; the step's first such lane keeps the count).  Longer
    // shares are counted first and ranked in a second pass over the points.
    constexpr uint32_t kKeep = 4;
    const uint32_t share = ((len + waves - 1u) / waves + 63u) & ~63u, begin = wave * share, end = min(len, begin + share);
    const bool kept = share <= 64u * kKeep;   // (uniform over the team)
    const uint64_t lanes_below = (1ull << lane) - 1ull;
    auto same_voxel = [&](uint32_t v, bool live) {   // lanes of this wave whose point lies in the same voxel
        uint64_t same = __ballot(live);
#pragma unroll
        for (uint32_t bit = 0; bit < 6; bit++) { const bool b = (v >> bit) & 1u; const uint64_t vote = __ballot(b); same &= b ? vote : ~vote; }
        return same;
    };
    for (uint32_t k = tid; k < waves * 64u; k += stride) (&s.run[0][0])[k] = 0u;
    sync();
    float4 q[kKeep];
    uint32_t vr[kKeep];   // voxel | rank among the share's points of that voxel << 8
    if (kept) {
#pragma unroll
        for (uint32_t j = 0; j < kKeep; j++) { const uint32_t k = begin + 64u * j + lane; q[j] = k < end ? pts[first + k] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (uint32_t j = 0; j < kKeep; j++) {
            vr[j] = 0u;
            if (begin + 64u * j >= end) continue;   // (wave-uniform) a step without points: the average block of a room scan has two of these
            const bool live = begin + 64u * j + lane < end;
            const uint32_t v = live ? voxel_in_block(q[j], recip) : 0u;
            const uint64_t same = same_voxel(v, live);
            if (live) {
                const uint32_t leader = uint32_t(__ffsll((unsigned long long)same)) - 1u;
                uint32_t base = 0;
                if (lane == leader) { base = s.run[wave][v]; s.run[wave][v] = base + uint32_t(__popcll(same)); }
                vr[j] = v | ((uint32_t(__shfl(base, int(leader))) + uint32_t(__popcll(same & lanes_below))) << 8);
            }
            wave_lds_fence();   // the next step's leaders read what this step's leaders wrote
        }
    } else
        for (uint32_t k = begin + lane; k < end; k += 64u) atomicAdd(&s.run[wave][voxel_in_block(pts[first + k], recip)], 1u);
    sync();
    // lane l holds voxel 63 - l, the order the points go in.  before = points in higher voxels, upto = before + own.
    uint32_t own = 0, mine_before = 0;   // points of this lane's voxel in the whole block / in the shares of lower waves
#pragma unroll
    for (uint32_t w = 0; w < uint32_t(kNormalWaves); w++)
        if (w < waves) { const uint32_t c = s.run[w][63u - lane]; if (w < wave) mine_before += c; own += c; }
    const uint32_t upto = wave_inclusive_sum(own);
    const uint32_t before = upto - own;
    sync();                                   // (every wave has read the counts)
    s.run[wave][63u - lane] = before + mine_before;   // first position of this voxel's points from this wave's share
    sync();
    // every point to its position -- the staged copy in LDS for the first kStagePoints positions, and the frame's ordered point
    // array (what the update kernel reads)
    if (kept) {
#pragma unroll
        for (uint32_t j = 0; j < kKeep; j++)
            if (begin + 64u * j + lane < end) {
                const uint32_t at = s.run[wave][vr[j] & 63u] + (vr[j] >> 8);
                ordered[first + at] = q[j];
                if (at < kStagePoints) s.stage[at] = make_float4(q[j].x, q[j].y, q[j].z, __uint_as_float(vr[j] & 63u));   // (the staged copy keeps the voxel, not the input index)
            }
    } else
        for (uint32_t k0 = begin; k0 < end; k0 += 64u) {
            const uint32_t k = k0 + lane;
            const bool live = k < end;
            const float4 p = live ? pts[first + k] : make_float4(0, 0, 0, 0);
            const uint32_t v = live ? voxel_in_block(p, recip) : 0u;
            const uint64_t same = same_voxel(v, live);
            if (live) {
                const uint32_t leader = uint32_t(__ffsll((unsigned long long)same)) - 1u;
                uint32_t base = 0;
                if (lane == leader) { base = s.run[wave][v]; s.run[wave][v] = base + uint32_t(__popcll(same)); }
                const uint32_t at = uint32_t(__shfl(base, int(leader))) + uint32_t(__popcll(same & lanes_below));
                ordered[first + at] = p;
                if (at < kStagePoints) s.stage[at] = make_float4(p.x, p.y, p.z, __uint_as_float(v));
            }
            wave_lds_fence();
        }
    if (len > kStagePoints) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // positions beyond the staged ones are read back from `ordered` below (by every wave of a team)
    sync();

    // greedy neighbourhood walk (normals.hpp:86-138) over the voxel runs, in registers.  The last point (if this block holds it)
    // sits in the lowest occupied voxel: that voxel counts one point less.
    const uint64_t occupied = __ballot(own != 0u);
    const uint32_t lowest = occupied ? 63u - uint32_t(__clzll((long long)occupied)) : 0u;   // highest lane with points = lowest voxel
    const uint32_t cnt = own - ((holds_last && lane == lowest && own != 0u) ? 1u : 0u);
    const uint32_t cupto = upto - ((holds_last && occupied != 0ull && lane >= lowest) ? 1u : 0u);   // (the scan of cnt: one less from the lowest voxel on)
    const uint32_t cbefore = cupto - cnt;     // (== before everywhere: the missing point is the very last one)
    // The walk visits the occupied voxels in descending order (ascending lanes); an octant is the eight lanes l & ~7 .. l | 7.  What it does,
    // said without the loop (the loop cost ~35 instructions per neighbourhood, a dozen neighbourhoods per block; checked against it for
    // 200 000 random count vectors): a voxel of at least kMinPoints points in front of its octant's first smaller one is a neighbourhood
    // of its own; that first smaller voxel opens the octant's TAIL -- itself and every voxel behind it in the octant; the first octant
    // whose tail stays below kMinPoints points takes the rest of the block with it and ends the walk.
    const uint64_t occupied_now = (ablate & 64u) ? 0ull : __ballot(cnt != 0u);
    const uint64_t large = __ballot(cnt >= kMinPoints) & occupied_now, small = occupied_now & ~large;
    const uint32_t oct0 = lane & ~7u;
    const uint32_t small_here = uint32_t(small >> oct0) & 0xffu;
    const uint32_t tail_lane = small_here ? oct0 + uint32_t(__ffs(int(small_here))) - 1u : 64u;   // first smaller voxel of this lane's octant (64: none)
    const uint32_t tail_size = uint32_t(__shfl(int(cupto), int(oct0 | 7u))) - uint32_t(__shfl(int(cbefore), int(tail_lane & 63u)));   // (meaningless without a tail)
    const uint64_t ending = __ballot(tail_lane != 64u && tail_size < kMinPoints);   // octants whose tail takes the rest of the block
    const uint32_t last_start = ending ? read_lane(tail_lane, uint32_t(__ffsll((unsigned long long)ending)) - 1u) : 64u;   // where the walk's last neighbourhood begins
    const uint64_t walked = last_start >= 63u ? ~0ull : (2ull << last_start) - 1ull;     // lanes the walk starts neighbourhoods in
    const bool alone = ((large >> lane) & 1ull) != 0ull && lane < tail_lane;
    const uint64_t starts = __ballot(alone || lane == tail_lane) & walked;
    const bool starter = ((starts >> lane) & 1ull) != 0ull;
    const uint32_t my_size = alone ? cnt : (lane == last_start ? read_lane(cupto, 63u) - cbefore : tail_size);
    const bool narrow = starter && my_size >= kMinPoints && my_size < kTeamPoints, wide = starter && my_size >= kTeamPoints && my_size < kCoopPoints;
    const uint64_t narrow_starts = __ballot(narrow), wide_starts = __ballot(wide);
    const uint32_t ngroups = uint32_t(__popcll(starts)), nnarrow = uint32_t(__popcll(narrow_starts)), nwide = uint32_t(__popcll(wide_starts));
    const uint32_t ncoop = uint32_t(__popcll(__ballot(starter && my_size >= kCoopPoints)));
    // the neighbourhood a voxel belongs to: the one that starts at the last neighbourhood's lane, at its octant's tail or at itself
    const uint32_t cover = (last_start != 64u && lane >= last_start) ? last_start : (tail_lane != 64u && lane >= tail_lane) ? tail_lane : lane;
    const uint32_t my_group = uint32_t(__popcll(starts & ((1ull << cover) - 1ull)));
    if (starter && wave == 0) {   // (for a starter, cover == lane: my_group is its own number)
        s.g_first[my_group] = cbefore; s.g_size[my_group] = my_size;
        if (narrow) s.team_list[__popcll(narrow_starts & lanes_below)] = my_group;
        if (wide) s.team_list[63u - uint32_t(__popcll(wide_starts & lanes_below))] = my_group;
    }
    if (wave == 0) s.group_of[63 - lane] = my_group;
    sync();
    const float4* __restrict__ sorted = ordered + first;   // the block in order (written above; positions below kStagePoints are in LDS too)

    // large neighbourhoods: a whole wave each, one after the other (dealt over the waves of a team)
    if (ncoop && !(ablate & 8u)) {
        uint32_t turn = 0;
        for (uint32_t g = 0; g < ngroups; g++) {
            const uint32_t g0 = s.g_first[g], gn = s.g_size[g];
            if (gn < kCoopPoints) continue;
            if ((turn++ % waves) != wave) continue;
            F3 nrm = g0 + gn <= kStagePoints ? fit_plane_wave(&s.stage[g0], gn, lane) : fit_plane_wave(sorted + g0, gn, lane);
            if (lane == 0) {
                const float4 q = g0 < kStagePoints ? s.stage[g0] : sorted[g0];
                const F3 to_sensor = unit3(sub3(pos, F3{q.x, q.y, q.z}));  // normals.hpp:117
                if (dot3(nrm, to_sensor) < 0.0f) nrm = {-nrm.x, -nrm.y, -nrm.z};
                s.g_normal[g][0] = nrm.x; s.g_normal[g][1] = nrm.y; s.g_normal[g][2] = nrm.z; s.g_normal[g][3] = 1.0f;
            }
        }
    }
    // Every other neighbourhood of at least kMinPoints points: teams of lanes -- eight lanes for 8..31 points (eight neighbourhoods per
    // wave at a time), sixteen for 32..255 (four at a time).  A team strides over its points (from the staged LDS copy, beyond it from
    // HBM) and combines the moments of the reference's two passes (normals.hpp:12-36) with DPP reductions; the covariance terms are
    // parked in LDS -- where the counting sort's counters were -- and the planes (normals.hpp:39-79) are worked out for all parked
    // neighbourhoods at once, one lane each: per round of teams that part cost as much again as the moments.  (Round 1 fitted the
    // neighbourhoods below 16 points with one lane each, in the reference's summation order: a handful of lanes busy for 700
    // instructions in three blocks out of four.)
    double (*cov)[6] = kTeam ? reinterpret_cast<double (*)[6]>(&(&s)[1]) : s.cov;   // a team of waves: the next wave's scratch is idle
    uint32_t* cov_group = kTeam ? reinterpret_cast<uint32_t*>(&(&s)[2]) : s.cov_group;
    const uint32_t cov_slots = kTeam ? 64u : kCovSlots;
    uint32_t pending = 0;   // parked neighbourhoods (uniform over the workgroup)
    auto planes = [&]() {
        sync();
        if (tid < pending && !(ablate & 16u)) {
            const uint32_t g = cov_group[tid], g0 = s.g_first[g];
            F3 nrm = plane_from_covariance(cov[tid][0], cov[tid][1], cov[tid][2], cov[tid][3], cov[tid][4], cov[tid][5]);
            const float4 q = g0 < kStagePoints ? s.stage[g0] : sorted[g0];
            const F3 to_sensor = unit3(sub3(pos, F3{q.x, q.y, q.z}));  // normals.hpp:117
            if (dot3(nrm, to_sensor) < 0.0f) nrm = {-nrm.x, -nrm.y, -nrm.z};
            s.g_normal[g][0] = nrm.x; s.g_normal[g][1] = nrm.y; s.g_normal[g][2] = nrm.z; s.g_normal[g][3] = 1.0f;
        }
        sync();
        pending = 0;
    };
    auto team_rounds = [&](auto width, uint32_t count, auto entry) {
        constexpr uint32_t kWidth = decltype(width)::value, kTeams = 64u / kWidth;
        const uint32_t team = lane / kWidth, tl = lane % kWidth;
        for (uint32_t done = 0; done < count; done += kTeams * waves) {
            const uint32_t round = min(kTeams * waves, count - done);   // neighbourhoods of this round, over all waves
            if (pending + round > cov_slots) planes();
            const uint32_t slot = wave * kTeams + team;
            const bool on = slot < round;
            const uint32_t g = on ? entry(done + slot) : 0u;
            const uint32_t g0 = s.g_first[g], gn = on ? s.g_size[g] : 0u;
            auto point = [&](uint32_t k) { const uint32_t at = g0 + k; return at < kStagePoints ? s.stage[at] : sorted[at]; };
            double cx = 0, cy = 0, cz = 0;
            for (uint32_t k = tl; k < gn; k += kWidth) { const float4 q = point(k); cx += double(q.x); cy += double(q.y); cz += double(q.z); }
            const double rcp = 1.0 / double(gn);
            cx = team_sum<kWidth>(cx) * rcp; cy = team_sum<kWidth>(cy) * rcp; cz = team_sum<kWidth>(cz) * rcp;
            double xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
            for (uint32_t k = tl; k < gn; k += kWidth) {
                const float4 q = point(k);
                const double rx = double(q.x) - cx, ry = double(q.y) - cy, rz = double(q.z) - cz;
                xx += rx * rx; xy += rx * ry; xz += rx * rz; yy += ry * ry; yz += ry * rz; zz += rz * rz;
            }
            xx = team_sum<kWidth>(xx); xy = team_sum<kWidth>(xy); xz = team_sum<kWidth>(xz);
            yy = team_sum<kWidth>(yy); yz = team_sum<kWidth>(yz); zz = team_sum<kWidth>(zz);
            if (on && tl == 0) {
                double* c = cov[pending + slot];
                c[0] = xx * rcp; c[1] = xy * rcp; c[2] = xz * rcp; c[3] = yy * rcp; c[4] = yz * rcp; c[5] = zz * rcp;
                cov_group[pending + slot] = g;
            }
            pending += round;
        }
    };
    if (!(ablate & 8u)) {
        team_rounds(std::integral_constant<uint32_t, 8>{}, nnarrow, [&](uint32_t i) { return s.team_list[i]; });
        team_rounds(std::integral_constant<uint32_t, 16>{}, nwide, [&](uint32_t i) { return s.team_list[63u - i]; });
        if (pending) planes();
    }
    // neighbourhoods below kMinPoints points: every point its own fallback normal (normals.hpp:127-134)
    if (tid < ngroups && s.g_size[tid] < kMinPoints) s.g_normal[tid][3] = 0.0f;
    sync();
    // every point picks up its neighbourhood's normal.  Two loops, so that the common one -- the staged points, whose LDS copy
    // carries the voxel -- does not pay for the other's floor / convert / interleave
    auto emit = [&](uint32_t k, float4 q, uint32_t voxel) {
        F3 nrm;
        bool fitted = false;
        if (k < grouped) {
            const uint32_t g = s.group_of[voxel];
            fitted = s.g_normal[g][3] != 0.0f;
            nrm = {s.g_normal[g][0], s.g_normal[g][1], s.g_normal[g][2]};
        }
        if (!fitted) nrm = unit3(sub3(pos, F3{q.x, q.y, q.z}));  // normals.hpp:130
        normals[first + k] = make_float4(nrm.x, nrm.y, nrm.z, 0.0f);
    };
    const uint32_t out_len = (ablate & 32u) ? 0u : len;
    for (uint32_t k = tid; k < min(out_len, kStagePoints); k += stride) { const float4 q = s.stage[k]; emit(k, q, __float_as_uint(q.w)); }
    for (uint32_t k = kStagePoints + tid; k < out_len; k += stride) { const float4 q = sorted[k]; emit(k, q, voxel_in_block(q, recip)); }
    sync();
}

// The update kernel's view of the work-item lists (FrameHeader::item_first / frame_sparse / first_item), by one workgroup.
__device__ __forceinline__ void publish_item_summary(FrameHeader* __restrict__ h, const uint4* __restrict__ items, uint32_t list_stride, bool sparse) {
    __shared__ uint32_t first[2][kListStripes + 1];
    if (threadIdx.x < 2 * kListStripes) {   // lanes 0..15: long items, 16..31: short items; a 16-lane inclusive scan each
        const uint32_t list = threadIdx.x / kListStripes, stripe = threadIdx.x % kListStripes;
        uint32_t incl = h->list_count[list ? kListShortItems : kListLongItems][stripe * 16];
#pragma unroll
        for (int o = 1; o < int(kListStripes); o <<= 1) { const uint32_t t = __shfl_up(incl, o, kListStripes); if (int(stripe) >= o) incl += t; }
        first[list][stripe + 1] = incl;
        if (stripe == 0) first[list][0] = 0u;
    }
    __syncthreads();
    if (threadIdx.x < 2 * (kListStripes + 1)) (&h->item_first[0][0])[threadIdx.x] = (&first[0][0])[threadIdx.x];
    if (threadIdx.x == 0) h->frame_sparse = sparse ? 1u : 0u;
    const uint32_t n_long = first[0][kListStripes], nitems = n_long + first[1][kListStripes];
    for (uint32_t v = threadIdx.x; v < min(nitems, kFirstItems); v += blockDim.x) {
        // long items fill a stripe's range of items[] from the front, short ones from the back (k_frame_gather)
        const bool is_long = v < n_long;
        const uint32_t* f = first[is_long ? 0 : 1];
        const uint32_t i = is_long ? v : v - n_long;
        uint32_t stripe = 0;
#pragma unroll
        for (uint32_t k = 1; k < kListStripes; k++) stripe = i >= f[k] ? k : stripe;
        const uint32_t index = i - f[stripe];
        h->first_item[v] = items[is_long ? stripe * list_stride + index : (stripe + 1u) * list_stride - 1u - index];
    }
}

// Workgroups of four waves.  In a dense frame the first kNormalBigGroups workgroups take the large point blocks (more than
// kBigBlock points: k_frame_gather lists them), one block at a time with all four waves; every other wave takes small blocks, one
// wave per block, the waves of a workgroup sharing nothing.  (The large workgroups have the lowest indices, so they start first:
// they are the kernel's long pole.)
__global__ void __launch_bounds__(64 * kNormalWaves, 5)  // <= 96 VGPRs: 20 waves per CU (7.2 KiB of LDS per wave)
k_frame_normals(float4* __restrict__ points0, float4* __restrict__ points1, float recip,
                const uint2* __restrict__ segments, uint32_t list_stride,
                FrameHeader* __restrict__ h, F3 pos, float4* __restrict__ normals, uint32_t ablate, uint32_t big_block, uint32_t max_big_groups,
                const uint4* __restrict__ items) {
    __shared__ NormalScratch scratch[kNormalWaves];
    const float4* __restrict__ pts = (h->sort_passes & 1u) ? points1 : points0;   // where the sort's last pass left the points: by block, input order inside
    float4* __restrict__ ordered = (h->sort_passes & 1u) ? points0 : points1;     // the sort's other buffer: the same blocks, each in the reference's order
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // where each stripe's entries begin in the two block lists (all blocks from the front of a stripe, the large ones again from its
    // back), in LDS: kept in registers the thirty-four prefix sums spill
    __shared__ uint32_t list_begin[2][kListStripes + 1];
    if (threadIdx.x < 2 * kListStripes) {   // lanes 0..15: all blocks, 16..31: the large ones; a 16-lane inclusive scan each
        const uint32_t which = threadIdx.x / kListStripes, stripe = threadIdx.x % kListStripes;
        uint32_t incl = h->list_count[which ? kListBigBlocks : kListBlocks][stripe * 16];
#pragma unroll
        for (int o = 1; o < int(kListStripes); o <<= 1) { const uint32_t t = __shfl_up(incl, o, kListStripes); if (int(stripe) >= o) incl += t; }
        list_begin[which][stripe + 1] = incl;
        if (stripe == 0) list_begin[which][0] = 0u;
    }
    __syncthreads();
    const uint32_t n = h->n_usable, nseg = list_begin[0][kListStripes], nbig = list_begin[1][kListStripes];
    auto locate = [&](uint32_t which, uint32_t i, uint32_t* index) {   // stripe of entry i of a list and its index inside the stripe
        uint32_t stripe = 0;
#pragma unroll
        for (uint32_t k = 1; k < kListStripes; k++) stripe = i >= list_begin[which][k] ? k : stripe;
        *index = i - list_begin[which][stripe];
        return stripe;
    };
    auto segment = [&](uint32_t i) {   // point block i, counted through the stripes of the list
        uint32_t index;
        const uint32_t stripe = locate(0u, i, &index);
        return segments[stripe * list_stride + index];
    };
    const bool sparse = n < kSparsePointsPerBlock * nseg;   // frame_is_sparse
    if (blockIdx.x == gridDim.x - 1u) publish_item_summary(h, items, list_stride, sparse);   // (never one of the large blocks' workgroups)
    // the large blocks' workgroups (none in a sparse frame or when there is no large block)
    uint32_t big_groups = 0;
    if (!sparse && !(ablate & 128u)) {
        big_groups = nbig ? min(min(max_big_groups, nbig), gridDim.x / 2u) : 0u;
        if (blockIdx.x < big_groups) {
            for (uint32_t b = blockIdx.x; b < nbig; b += big_groups) {
                uint32_t index;
                const uint32_t stripe = locate(1u, b, &index);
                estimate_block<true>(scratch[0], segments[(stripe + 1u) * list_stride - 1u - index], pts, ordered, recip, pos, normals, ablate);
            }
            return;
        }
    }
    NormalScratch& s = scratch[wave];
    const uint32_t my_wave = (blockIdx.x - big_groups) * kNormalWaves + wave, all_waves = (gridDim.x - big_groups) * kNormalWaves;
    if (!sparse) {  // dense frame: one wave per (small) point block
        uint2 next = my_wave < nseg ? segment(my_wave) : make_uint2(0u, 0u);
        for (uint32_t seg = my_wave; seg < nseg; seg += all_waves) {
            const uint2 record = next;
            if (seg + all_waves < nseg) next = segment(seg + all_waves);   // the next block's record: in flight while this one is worked off
            if (big_groups && (record.y & 0x7fffffffu) > big_block) continue;   // a large block: the workgroups above
            estimate_block<false>(s, record, pts, ordered, recip, pos, normals, ablate);
        }
        return;
    }
    // Sparse frame (about one point per block, e.g. 2 cm voxels outdoors): a block with fewer than kMinPoints points cannot
    // form a neighbourhood (normals.hpp:88: the walk never leaves the block), so all its points get the per-point
    // fallback normal -- one LANE per such block, 64 blocks per wave step, instead of a wave per block (334 k blocks
    // per cfg5 frame: 167 us).  The few larger blocks go through the wave path afterwards.
    // Batch t holds the blocks lane * batches + t: the larger blocks cluster (next to the sensor), and sixty-four
    // neighbours in one batch would be worked off one after the other by a single wave.
    const uint32_t batches = (nseg + 63u) / 64u;
    for (uint32_t t = my_wave; t < batches; t += all_waves) {
        const uint32_t mine = lane * batches + t;
        const uint2 record = mine < nseg ? segment(mine) : make_uint2(0u, 0u);
        const uint32_t len = record.y & 0x7fffffffu;
        const bool small = mine < nseg && len != 0u && len < kMinPoints;
        if (small)   // the block's points into the reference's order (descending voxel, ties in input order = arrival order) by counting, then the fallback normal
            for (uint32_t k = 0; k < len; k++) {
                const float4 q = pts[record.x + k];
                const uint32_t v = voxel_in_block(q, recip);
                uint32_t at = 0;
                for (uint32_t j = 0; j < len; j++) {
                    const uint32_t u = voxel_in_block(pts[record.x + j], recip);
                    at += (u > v || (u == v && j < k)) ? 1u : 0u;
                }
                const F3 nrm = unit3(sub3(pos, F3{q.x, q.y, q.z}));  // normals.hpp:130
                ordered[record.x + at] = q;
                normals[record.x + at] = make_float4(nrm.x, nrm.y, nrm.z, 0.0f);
            }
        for (uint64_t big = __ballot(mine < nseg && !small); big; big &= big - 1ull) {
            const uint32_t l = uint32_t(__ffsll((unsigned long long)big)) - 1u;
            estimate_block<false>(s, make_uint2(read_lane(record.x, l), read_lane(record.y, l)), pts, ordered, recip, pos, normals, ablate);
        }
    }
}

}  // namespace chadhip

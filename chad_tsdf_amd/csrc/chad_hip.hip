// chad_hip.hip -- host side of the C ABI in include/chad_hip.h: owns the device-resident map, applies the
// reference's submap roll-over rule (tsdf.cpp:46-61) and enqueues the per-frame kernel pipeline
// (tsdf.cpp:64-70: calc_morton_vector -> sort_morton_vector -> estimate_normals -> Octree::insert).
// There is no CPU fallback: without a HIP device every entry point fails with CHAD_ERR_NO_DEVICE.
#include "../../include/chad_hip.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "alloc.hpp"
#include "comm.hpp"
#include "dag.hpp"
#include "dag_device.hpp"
#include "exchange_kernels.hpp"
#include "frame_kernels.hpp"
#include "map_kernels.hpp"
#include "merge_kernels.hpp"
#include "mesh_kernels.hpp"
#include "query_kernels.hpp"
#include "radix_sort.hpp"
#include "staging.hpp"
#include "tile_update.hpp"

using namespace chadhip;

namespace {

thread_local std::string g_error;

chad_status fail(chad_status code, const std::string& msg) {
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            (void)hipGetLastError();                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? CHAD_ERR_OUT_OF_MEMORY                                     \
                        : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorNoBinaryForGpu || e_ == hipErrorInsufficientDriver) ? CHAD_ERR_NO_DEVICE \
                                                                                                               : CHAD_ERR_HIP,                                             \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                    \
        }                                                                                                      \
    } while (0)

#define TRY(expr)                          \
    do {                                   \
        chad_status s_ = (expr);           \
        if (s_ != CHAD_OK) return s_;      \
    } while (0)

struct Submap {  // a closed submap: dense copies of its bricks (stands in for the reference's DAG, out of scope)
    uint64_t* keys = nullptr;
    float* sum = nullptr;
    uint32_t* cnt = nullptr;
    uint32_t blocks = 0;
    bool released = false;  // CHAD_FLAG_DAG_ONLY: folded into the DAG, the dense copy given back
    bool borrowed = false;  // its arrays are carved out of chad_map::archive_arena (CHAD_FLAG_DAG_ONLY: the copy lives from the roll-over to the fold), not allocations of their own
};

struct DevBuf {   // device allocation released on every exit path
    void* p = nullptr;
    ~DevBuf() { if (p) (void)chad_free(p); }
    hipError_t alloc(size_t bytes) { return chad_malloc(&p, bytes ? bytes : 8); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

struct Arena {   // grow-only device scratch, carved up again for every use (no hipMalloc / hipFree -- each a device-wide stall -- in steady state)
    char* base = nullptr;
    size_t cap = 0, used = 0;
    hipError_t reserve(size_t bytes) {   // previous contents are dead
        used = 0;
        if (bytes <= cap) return hipSuccess;
        if (base) (void)chad_free(base);
        base = nullptr; cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        const hipError_t e = chad_malloc(reinterpret_cast<void**>(&base), want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    template <class T> T* take(size_t count) {
        used = (used + 255) & ~size_t(255);
        T* p = reinterpret_cast<T*>(base + used);
        used += count * sizeof(T);
        return p;
    }
    static size_t padded(size_t bytes) { return ((bytes + 255) & ~size_t(255)) + 256; }
    void release() { if (base) (void)chad_free(base); base = nullptr; cap = used = 0; }
};

enum Stage { T_BEGIN = 0, T_CALC, T_SORT, T_NORM, T_UPD, T_COUNT };

}  // namespace

// Everything one frame needs on its way from raw points to (points, normals, work items) in processing order.  A few
// slots, used in turn, each with a stream of its own, on which the whole frame runs: preparation (bounds, keys, sort,
// gather, normals -- none of which touches the map; every kernel of it is latency-bound and leaves most of the chip
// idle) and then the update.  Updates of different frames commute (sums and counts, added atomically), so the streams
// need no ordering among themselves.  What does need the map to itself -- clearing it, a roll-over, growth, an export --
// waits for the updates enqueued so far and makes the later ones wait for it (exclusive_begin / exclusive_end below).
struct FrameSlot {
    hipStream_t prep = nullptr;
    hipEvent_t update_done = nullptr;   // recorded on `prep` behind its last update when somebody has to wait for it (updates_done)
    bool tail_unrecorded = false;       // frames have been enqueued on `prep` since update_done was last recorded
    bool used = false;
    uint64_t seen_epoch = 0;            // the last exclusive operation this stream has been made to wait for
    FrameHeader* d_hdr = nullptr;
    FramePartial* d_part = nullptr;     // kFrameBlocks partial reductions of the frame kernels
    size_t cap_points = 0;
    float* d_xyz = nullptr;             // device copy of a host frame
    float* h_stage = nullptr;           // pinned staging buffer for chad_map_insert (host points), cap_points * 12 bytes
    hipEvent_t stage_done = nullptr;    // the H2D copies out of h_stage have completed
    uint32_t input_seq = 0;             // frames enqueued on this slot so far; the device stores the number into chad_map::h_consumed[slot]
                                        // once the frame's input (the caller's points) has been read for the last time
    RadixSortBuffers sort{};
    float4* d_normals = nullptr;
    uint32_t* d_seg = nullptr;
};

struct chad_iter;
struct chad_map {
    float res = 0, trunc = 0, recip = 0;
    int device = 0;
    hipStream_t stream = nullptr;       // the map's stream: every operation that needs the map to itself
    hipEvent_t excl_done = nullptr;     // recorded behind the last such operation
    uint64_t excl_epoch = 0;            // ... and counted
    bool pending_clear = false;         // chad_map_reset is lazy: the clear runs on the next insert's stream, right before its update
    bool pending_clear_overflow = false;
    bool own_stream = false;
    hipEvent_t caller_ready = nullptr;  // caller-supplied stream only: the frame's input is ready once this has passed
    hipEvent_t producer_ready = nullptr;  // chad_map_insert_device_stream: recorded on the producer's stream, waited for by the frame's stream
    volatile uint32_t* h_consumed = nullptr;  // host-mapped, one 64-byte line per slot: FrameSlot::input_seq of the last frame whose input the device has consumed
    uint32_t* d_consumed = nullptr;           // (its device address)
    RadixSortBuffers aux_sort{};        // block-key sorter of exports and of Submap::finalize (never a frame slot's: those belong to frames in flight)
    size_t aux_cap = 0;
    // host time of the insert path, cumulative (chad_map_host_times): in the insert calls altogether / of that, waiting for the device to
    // hand a frame slot back (prepare_slot: the bound of three frames in flight) / of that, in the H2D copy of chad_map_insert
    double host_insert_s = 0, host_wait_s = 0, host_copy_s = 0;
    uint64_t host_inserts = 0;
    uint32_t cu_count = 256;            // compute units of the map's device (hipDeviceProp_t::multiProcessorCount): every grid that is sized "per CU" uses it
    TileLaunch tile{};                  // launch shape of the update kernel on this device for this map's resolution / truncation (tile_update.hpp)
    uint32_t sort_resident = 0;         // workgroups of the frame sort's pass kernel the device certainly holds at once (radix_resident_workgroups)
    uint32_t aux_resident = 0;
    static constexpr uint32_t kMaxSlots = 8;
    uint32_t n_slots = 3;               // frames in flight (CHAD_SLOTS overrides, for experiments)
    FrameSlot slot[kMaxSlots];
    bool serial = false;                // stage timing / verbose: the slots share the map's stream, so that stage times do not overlap
    uint32_t last_slot = 0;             // slot of the last insert
    uint32_t sort_passes_forced = 0;    // CHAD_DEBUG_SORT_PASSES: single-pass sort launches per frame (tests of the guarded tail kernel)
    uint32_t flags = 0, update_mode = CHAD_UPDATE_AUTO;
    uint32_t ablate = 0;  // CHAD_DEBUG_ABLATE: timing experiments on the normals kernel only (scripts/ablate_normals.sh)
    unsigned long long* d_stamps = nullptr;  // CHAD_DEBUG_STAMPS: per-phase cycle stamps of the update kernel

    MapView view{};
    uint32_t hash_capacity = 0, brick_capacity = 0;
    FrameHeader* h_hdr = nullptr;  // pinned
    uint32_t* d_overflow = nullptr;  // 16 words.  Word 0: sticky fault bits of the frame sorts (kSortFaultStalled); words 1..7: MapView::overflow of the tables
                                     // alive (active, spare, closing: each table has a word of its own, so that clearing one table -- ordered behind ITS updates --
                                     // cannot wipe what a frame in flight reports about another); word 8: launches that have used the scatter form's staging words
    uint32_t overflow_words_used = 0;   // bit w: word w belongs to a table
    uint32_t overflow_latched = 0;      // overflow bits of tables that have been closed since the last reset (sticky, like the active table's own)
    static constexpr uint32_t kRecentFrames = 2 * 8;   // two per frame slot at most (fold_if_due)
    size_t recent_points[kRecentFrames] = {};
    uint32_t recent_at = 0;
    uint64_t pack_points = 0;        // points inserted since the last fold of the staging words was considered
    uint32_t pack_seen_folded = 0;   // what the device's "staging words in use" counter showed then
    uint64_t folds = 0;
    uint64_t fold_due = pack_fold_points(3);   // points since the last fold that make the next one due (set for the map's slots at creation; CHAD_DEBUG_FOLD_POINTS: tests fold often)
    bool report_folds = false;
    volatile uint32_t* h_live_blocks = nullptr;  // host-mapped word the device mirrors the live block count into each frame
    uint32_t* d_live_blocks = nullptr;           // (its device address): lets the host grow the pools without a round trip
    bool fixed_capacity = false;


    std::vector<std::array<float, 3>> positions;  // poses of the active submap (submap.hpp:110)
    std::vector<std::vector<std::array<float, 3>>> closed_positions;   // ... and of the closed ones, in closing order (what a pose-graph back-end needs)
    std::vector<std::array<float, 12>> closed_pose;                    // rigid correction per closed submap (rotation row-major, translation); empty = identity
    std::vector<Submap> closed;                   // tsdf.cpp:54
    Submap active_copy;                           // dense snapshot of the active submap made for the last export call
    Dag dag;                                      // tsdf.cpp:30 (_node_levels_p): one DAG for all submaps, built lazily (chad_map_finalize).  With the fold on the
                                                  // device (the default) this is the host's MIRROR of the level arrays, brought up to date by dag_host()
    DeviceDag ddag;                               // ... the stores and their look-up tables themselves (dag_device.hpp)
    DagFold fold;                                 // the fold in flight (at most one): closed submap number dag_roots.size()
    bool folding = false;
    bool in_fold_setup = false;                   // close_into_dag is setting the fold up (it uses the auxiliary sorter itself)
    hipStream_t fold_stream = nullptr;            // CHAD_FLAG_DAG_ONLY roll-overs: the fold into the DAG runs here, beside the frames and beside the map's
    hipStream_t close_stream = nullptr;           // ... and the step before it -- the closing table's keys, their order, its cluster words -- here: the next
                                                  // submap's, beside the fold of the one before (a fold takes as long as the eleven frames of a cfg5 submap and longer)
    struct PendingFold {                          // a closed submap whose cluster words are made (or on their way, close_stream) and whose fold waits for the one in flight
        bool set = false;
        const uint64_t* keys = nullptr; const uint64_t* tsdf = nullptr; const uint64_t* weight = nullptr; const uint8_t* mask = nullptr;
        uint32_t blocks = 0, parity = 0;
    } pending;
    hipEvent_t clusters_ready = nullptr;          // behind the pending submap's cluster words on close_stream
    uint32_t closings = 0;                        // roll-overs closed into the DAG so far: closing k's cluster words live in cluster_arenas[k & 1]
                                                  // stream (whose next roll-over must not queue behind a fold of several milliseconds)
    hipEvent_t fold_t0 = nullptr, fold_t1 = nullptr;   // device time of the fold in flight (step B of a roll-over)
    hipEvent_t close_t0[2] = {nullptr, nullptr}, close_t1[2] = {nullptr, nullptr};   // ... and of step A (keys, order, cluster words) of closing k, by k & 1
    uint32_t fold_parity = 0;                     // which of the two the submap being folded is
    bool dag_broken = false;                      // a fold failed half-way (out of memory for a level's growth ...): the DAG is incomplete until the map is reset
    // Where a roll-over's time goes (chad_map_pipeline_times; cumulative, restarted by the call): host seconds inside the roll-over's bookkeeping
    // (close_active / finish_closing / close_into_dag), inside fold_advance (enqueuing a fold's stages), of both the part spent BLOCKED until a fold
    // in flight had finished; device milliseconds of steps A and B summed over the folds completed; counts.
    struct PipelineTimes {
        double rollover_host_s = 0, fold_host_s = 0, fold_wait_s = 0, step_a_ms = 0, step_b_ms = 0;
        uint32_t rollovers = 0, folds_completed = 0;
        uint64_t exchange_bytes = 0, exchanges = 0, exchange_union_blocks = 0;   // sharded maps: bytes this rank handed to the collectives of its exchanges / exchanges completed / blocks in the last union
    } pt;
    bool in_rollover_span = false;
    int debug_fail_fold = 0, debug_fail_close = 0;   // CHAD_DEBUG_FAIL_FOLD / _CLOSE = n: the n-th fold stage / the n-th roll-over's step A of this map fails as if out of memory (tests of the recovery paths)
    AllocMeter alloc_mark;                        // the calling thread's allocation meter at the last chad_map_pipeline_times
    uint64_t growths_mark = 0;
    int spare_flags = 0;                          // the spare's flag word still holds: 1 = its "exact rows" bit (cleared at a roll-over), 2 = everything (it was
                                                  // the active table when the map was reset); spare_ready() resets it before the table is swapped in
    bool dag_on_host = false;                     // CHAD_DEBUG_DAG_HOST: the sequential host fold of dag.hpp instead (what the tests hold the device fold against)
    bool dag_mirror_stale = false;
    std::vector<std::array<uint32_t, 2>> dag_roots;  // submap.hpp:108-109 of the closed submaps folded so far (a prefix of `closed`)
    float dag_ms_device = 0, dag_ms_host = 0;
    uint64_t dag_generation = 0;                  // bumped when the DAG is thrown away (chad_map_reset): iterators of an older generation are dead
    std::vector<chad_iter*> iterators;            // the iterators handed out and not yet destroyed: chad_map_destroy cuts them loose
    // the last mesh extracted (chad_map_mesh is called twice: sizes, then data)
    int32_t mesh_submap = INT32_MIN;
    std::vector<float> mesh_vertices, mesh_normals;   // (normals: per vertex, lvr2.cpp:296-297)
    std::vector<uint32_t> mesh_triangles;
    // Roll-over without a pipeline stall: the active table is swapped for an empty spare one of the same size and is
    // archived (counted, compacted into a dense closed submap, cleared) on the map's stream behind the frames, the host
    // part of it -- sizing the archive -- whenever the count has arrived.  At most one submap is closing at a time.
    MapView spare{};
    bool has_spare = false;
    bool spare_unaffordable = false;              // a second table did not fit once: resets clear in place from then on
    MapView closing_view{};
    bool closing = false;
    hipEvent_t closing_counted = nullptr;         // the closing table's block count has reached closing_count
    uint32_t* closing_count = nullptr;            // pinned: the counter stripes of the closing table
    uint32_t* d_cursor = nullptr;                 // output cursor of k_compact_bricks
    uint64_t frames = 0, last_points_in = 0;
    uint64_t frames_empty_tail = 0;               // empty inserts since the last one with points (they use no slot)
    // Frame-sharded multi-GPU map (chad_map_attach_comm): insert k of the stream is integrated by rank k % world only, and a
    // submap is closed by merging the ranks' partial bricks (merge_and_close) instead of archiving the local ones.
    int stage_mode = 0;                           // chad_map_insert (host points): 0 = direct copy for frames of stage_direct_from bytes and more, staged below; 1 = always staged; 2 = always direct
    size_t stage_direct_from = size_t(512) << 10; // (CHAD_STAGE_MODE = staged | direct, CHAD_STAGE_DIRECT_KB: read when the map is created)
    bool copy_pool_failed = false;
    std::unique_ptr<CopyPool> copy_pool;          // helper threads of chad_map_insert's staging copy (made by the first large host frame)
    chad_comm* comm = nullptr;
    uint64_t stream_frames = 0;                   // inserts seen since the communicator was attached
    // The exchange runs like the local archive: beside the next frames.  The closing table is this rank's share; three stages, each
    // enqueued on the map's stream when the previous one's numbers have reached the host (merge_advance): polled by every insert,
    // waited for by whatever needs the closed submap.
    chad_comm* closing_comm = nullptr;            // non-null: the closing table is being merged over this communicator
    bool in_exchange = false;                     // join_streams is driving the exchange (no re-entry from inside it)
    int merge_stage = 0;                          // 0: local block count pending, 1: all ranks' counts pending, 2: size of the union pending
    hipEvent_t merge_event = nullptr;             // behind the device-to-host copy of the current stage's numbers
    uint64_t* merge_host = nullptr;               // pinned: [0..15] blocks per rank, [16] blocks in the union, [17] this rank's count (source of an H2D copy)
    Arena merge_a, merge_b, merge_c;              // scratch of the three stages
    Arena archive_arena;                          // CHAD_FLAG_DAG_ONLY: the closing submap's dense copy, from its roll-over to its fold (borrow_submap)
    Arena cluster_arena;                          // the cluster words of the submap being folded (device_cluster_words)
    Arena cluster_arenas[2];                      // ... under CHAD_FLAG_DAG_ONLY: of the submap being folded and of the one waiting behind it
    // The SPARSE exchange (exchange_kernels.hpp; sharded maps under CHAD_FLAG_DAG_ONLY): what exists travels, ownership is by Morton key range,
    // every rank turns its range into cluster words and rank 0 folds the submap -- once.  Five stages (sparse_stages), polled like the dense ones.
    static constexpr uint32_t kMergeHostWords = 4096;   // merge_host: [0, 1024) heads of all ranks | [1024, 1040) splitters | [1056, 1568) the G x G matrix | [1600, 1616) union sizes | [1632] this rank's
    bool closing_sparse = false;
    hipEvent_t comm_wait_event = nullptr;         // the exchange's event being waited for, and since when (comm_wait: CHAD_COMM_TIMEOUT_S)
    std::chrono::steady_clock::time_point comm_wait_since{};
    bool dag_remote = false;                      // this rank's closed submaps were folded on rank 0: it holds no DAG of its own
    struct SparseExchange {
        uint32_t nb = 0, wide = 0, n_union = 0, recv_blocks = 0;
        uint64_t* keys_sorted = nullptr; uint32_t* slot_sorted = nullptr; uint64_t* mask = nullptr; uint32_t* prefix = nullptr; uint32_t* first_block = nullptr;
        uint32_t* chunk_sums = nullptr; uint32_t* d_words = nullptr;
        uint64_t* d_small = nullptr;              // head | all heads | splitters | row | matrix | union size | all union sizes (device side of merge_host's words)
        char* send = nullptr; char* recv = nullptr;
        uint64_t* sorted = nullptr; uint32_t* heads = nullptr; uint32_t* d_union_n = nullptr;
        uint64_t blocks_from[16] = {}, words_from[16] = {};   // what every rank sends to THIS rank
        uint32_t wide_of[16] = {};
        size_t recv_off[16] = {};
        char* out = nullptr; size_t out_bytes = 0;   // this range's result: keys | tsdf words | (weight words) | masks, contiguous
    } sx;
    uint32_t merge_blocks = 0;                    // this rank's blocks
    uint64_t* merge_keys = nullptr; long long* merge_sum = nullptr; uint32_t* merge_cnt = nullptr;   // ... compacted raw (arena a)
    uint64_t merge_total = 0;                     // blocks over all ranks
    uint32_t* merge_heads = nullptr;              // (arena b) per-chunk counts of distinct keys
    uint64_t* merge_sorted = nullptr;             // (arena b) all ranks' block keys, sorted: kept from stage 1 to stage 2 in a buffer of the exchange's own
                                                  // (aux_sort belongs to whoever sorts next: an export or a finalize between the two stages)
    // stage timing (CHAD_FLAG_TIMING): a ring of event sets, one per insert, read back lazily by chad_map_stats so
    // that timing adds no synchronisation to insert()
    static constexpr int kTimingRing = 64;
    hipEvent_t ev[kTimingRing][T_COUNT] = {};
    uint64_t timed_begin = 0, timed_end = 0;  // inserts [timed_begin, timed_end) have unread event sets
    float ms[T_COUNT] = {};                   // mean stage times over the inserts read at the last stats call
    // kernel timing (CHAD_FLAG_TIME_KERNELS): every kernel of an insert is launched with a pair of events that take the
    // dispatch's own begin / end time stamps (hipExtLaunchKernel); frames then run one after the other on one stream
    static constexpr int kKernelRing = 16;
    hipEvent_t kev[kKernelRing][2 * CHAD_KERNEL_COUNT] = {};
    uint32_t kev_launched[kKernelRing] = {};  // bit id: kernel `id` was launched for that insert
    uint64_t ktimed_begin = 0, ktimed_end = 0;
    double kernel_ms[CHAD_KERNEL_COUNT] = {};
    uint64_t kernel_launches[CHAD_KERNEL_COUNT] = {};
};

namespace {

struct HostSpan {   // host time of a scope into a counter of chad_map::pt
    double& into; std::chrono::steady_clock::time_point t0;
    explicit HostSpan(double& counter) : into(counter), t0(std::chrono::steady_clock::now()) {}
    ~HostSpan() { into += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

struct DeviceGuard {  // entry points run on the map's device and restore the caller's
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess)) ok = true;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

inline dim3 grid_for(size_t n, int threads = 256, int max_blocks = 2048) {
    size_t b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > size_t(max_blocks)) b = max_blocks;
    return dim3(unsigned(b));
}

void free_sort_buffers(RadixSortBuffers& b) {
    (void)chad_free(b.keys[0]); (void)chad_free(b.keys[1]); (void)chad_free(b.idx[0]); (void)chad_free(b.idx[1]);
    (void)chad_free(b.points[0]); (void)chad_free(b.points[1]);
    (void)chad_free(b.hist_part); (void)chad_free(b.digit_base); (void)chad_free(b.tile_status);
    const uint32_t resident = b.resident, rest_limit = b.rest_limit;
    b = RadixSortBuffers{};
    b.resident = resident; b.rest_limit = rest_limit;
}

void free_frame_buffers(FrameSlot& f) {
    if (f.h_stage) (void)hipHostFree(f.h_stage);
    f.h_stage = nullptr;
    (void)chad_free(f.d_xyz);
    free_sort_buffers(f.sort);
    (void)chad_free(f.d_normals); (void)chad_free(f.d_seg);
    f.d_xyz = nullptr; f.d_normals = nullptr; f.d_seg = nullptr;
    f.cap_points = 0;
}

// Wait for everything enqueued for this map, on all of its streams.  A multi-GPU exchange in progress is driven to its end first:
// its later stages are only enqueued when the host looks (merge_advance), and a host that blocks here -- in sync, stats, an export,
// a query, growth -- without looking would leave the peers' collectives waiting on the device for a stage this rank never issues.
// Every blocking entry point of the library comes through here (or through finish_closing(m, true) itself), so a rank that stops
// inserting right after a roll-over still completes its share as soon as it calls anything that waits.
chad_status finish_closing(chad_map* m, bool block);
chad_status join_streams(chad_map* m) {
    if (m->closing_comm && !m->in_exchange) {
        m->in_exchange = true;
        const chad_status s = finish_closing(m, true);
        m->in_exchange = false;
        if (s != CHAD_OK) return s;
    }
    for (auto& f : m->slot) if (f.prep && f.prep != m->stream) HIP_TRY(hipStreamSynchronize(f.prep));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->close_stream) HIP_TRY(hipStreamSynchronize(m->close_stream));
    if (m->fold_stream) HIP_TRY(hipStreamSynchronize(m->fold_stream));
    return CHAD_OK;
}

// The event behind the last update enqueued on a slot's stream.  Recorded here, when somebody is about to wait for it, and
// not once per frame: a stream of small frames is bound by the host's launch rate, and most frames are never waited for.
chad_status updates_done(FrameSlot& f, hipEvent_t* out) {
    if (f.tail_unrecorded) { HIP_TRY(hipEventRecord(f.update_done, f.prep)); f.tail_unrecorded = false; }
    *out = f.update_done;
    return CHAD_OK;
}

// The map stream is about to do something that needs the map to itself: it waits (on the device) for every update
// enqueued so far.
chad_status exclusive_begin(chad_map* m) {
    for (uint32_t k = 0; k < m->n_slots; k++) {
        FrameSlot& f = m->slot[k];
        if (!f.used || f.prep == m->stream) continue;
        hipEvent_t done;
        TRY(updates_done(f, &done));
        HIP_TRY(hipStreamWaitEvent(m->stream, done, 0));
    }
    return CHAD_OK;
}
// ... and is done with it: updates enqueued from now on wait for this point (insert_device checks the epoch).
chad_status exclusive_end(chad_map* m, hipStream_t stream) {
    HIP_TRY(hipEventRecord(m->excl_done, stream));
    m->excl_epoch++;
    return CHAD_OK;
}

chad_status ensure_frame_capacity(chad_map* m, FrameSlot& f, size_t n) {
    if (n <= f.cap_points) return CHAD_OK;
    TRY(join_streams(m));
    size_t cap = std::max<size_t>(n, f.cap_points * 2);
    cap = (cap + kSortTile - 1) / kSortTile * kSortTile;
    free_frame_buffers(f);
    HIP_TRY(chad_malloc(&f.d_xyz, cap * 12));
    for (int i = 0; i < 2; i++) {
        HIP_TRY(chad_malloc(&f.sort.keys[i], cap * 8));
        HIP_TRY(chad_malloc(&f.sort.points[i], cap * 16));  // the points travel with their keys; the last pass's output is the frame in processing order
    }
    HIP_TRY(chad_malloc(&f.sort.hist_part, size_t(kMaxHistBlocks) * kMaxPasses * kRadix * 4));
    HIP_TRY(chad_malloc(&f.sort.digit_base, size_t(kBaseStripes) * kMaxPasses * kRadixMax * 4));
    HIP_TRY(chad_malloc(&f.sort.tile_status, radix_status_words(radix_tiles(cap)) * 4));
    f.sort.tiles_capacity = radix_tiles(cap);
    HIP_TRY(chad_malloc(&f.d_normals, cap * 16));
    // point blocks uint2 (start, length | flag) and update work items uint4 (start, length, super-block code), each list in
    // kListStripes stripes of cap / kListStripes + kGatherChunk entries (FrameHeader::list_count)
    HIP_TRY(chad_malloc(&f.d_seg, (cap + size_t(kListStripes) * kGatherChunk) * 24));
    f.cap_points = cap;
    return CHAD_OK;
}

// The (key, index) sorter of exports and finalize: buffers of its own, so that no frame slot is ever borrowed (a slot's buffers
// belong to a frame that may still be in flight, and growing them would free a frame's staged points).  Used on the map's stream only.
chad_status fold_advance(chad_map* m, bool block);   // (with Submap::finalize, further down)
inline bool fold_busy(const chad_map* m) { return m->folding || m->pending.set; }   // a roll-over's fold in flight, or waiting for the one in flight
// the words of RadixSortSignals::dep in the auxiliary sorter: behind the partial histograms the histogram kernel's kHistBlocks workgroups write
inline uint32_t* aux_sort_dep(const RadixSortBuffers& b) { return b.hist_part + size_t(kHistBlocks) * kMaxPasses * kRadix; }
chad_status ensure_aux_sort(chad_map* m, size_t n) {
    if (fold_busy(m) && !m->in_fold_setup) {   // a roll-over in flight sorted with these buffers on its own stream: the map's stream may have them when it is done
        while (fold_busy(m)) TRY(fold_advance(m, true));
    }
    if (n <= m->aux_cap) return CHAD_OK;
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->close_stream) HIP_TRY(hipStreamSynchronize(m->close_stream));   // (the previous roll-over's sort)
    size_t cap = std::max<size_t>(n + n / 2, m->aux_cap * 2);   // (with room: the submaps of a trajectory differ by tens of per cent, and these seven allocations stall the device)
    cap = (cap + kSortTile - 1) / kSortTile * kSortTile;
    free_sort_buffers(m->aux_sort);
    m->aux_cap = 0;
    RadixSortBuffers& b = m->aux_sort;
    for (int i = 0; i < 2; i++) { HIP_TRY(chad_malloc(&b.keys[i], cap * 8)); HIP_TRY(chad_malloc(&b.idx[i], cap * 4)); }
    HIP_TRY(chad_malloc(&b.hist_part, size_t(kMaxHistBlocks) * kMaxPasses * kRadix * 4));
    HIP_TRY(chad_malloc(&b.digit_base, size_t(kBaseStripes) * kMaxPasses * kRadix * 4));
    HIP_TRY(chad_malloc(&b.tile_status, radix_status_words(radix_tiles(cap)) * 4));
    HIP_TRY(hipMemsetAsync(aux_sort_dep(b), 0, kDepWords * 4, m->stream));   // (RadixSortSignals::dep: zero before a sort; the sort leaves it so)
    HIP_TRY(hipStreamSynchronize(m->stream));
    b.tiles_capacity = radix_tiles(cap);
    b.resident = m->aux_resident;
    m->aux_cap = cap;
    return CHAD_OK;
}

void free_map_storage(chad_map* m, MapView& v) {
    (void)chad_free(v.hash_keys); (void)chad_free(v.live_bits);
    (void)chad_free(v.brick_sum); (void)chad_free(v.brick_cnt); (void)chad_free(v.brick_pack); (void)chad_free(v.brick_count);
    if (v.overflow) m->overflow_words_used &= ~(1u << uint32_t(v.overflow - m->d_overflow));
    v = MapView{};
}

// Hash table + brick pools for `blocks` bricks, empty (keys EMPTY, values PENDING, sums and counts zero).
chad_status alloc_map_storage(chad_map* m, MapView& v, uint64_t blocks, hipStream_t stream) {
    uint64_t hcap = 1024;
    while (hcap < blocks * 2) hcap <<= 1;  // load factor <= 0.5
    v = MapView{};
    uint32_t word = 1;
    while (word < 8 && ((m->overflow_words_used >> word) & 1u)) word++;
    if (word == 8) return fail(CHAD_ERR_HIP, "internal: more than seven tables alive");
    m->overflow_words_used |= 1u << word;
    v.overflow = m->d_overflow + word;
    HIP_TRY(hipMemsetAsync(v.overflow, 0, 4, stream));
    HIP_TRY(chad_malloc(&v.hash_keys, hcap * 8));
    HIP_TRY(chad_malloc(&v.live_bits, hcap / 8));
    HIP_TRY(chad_malloc(&v.brick_sum, hcap * 512));   // a brick lives at its hash slot; 64-bit sums (map_kernels.hpp)
    HIP_TRY(chad_malloc(&v.brick_cnt, hcap * 256));
    HIP_TRY(chad_malloc(&v.brick_pack, hcap * 512));  // staging words of the scatter form (map_kernels.hpp, kPackCountShift)
    HIP_TRY(chad_malloc(&v.brick_count, kCountStripes * 64));
    v.hash_mask = uint32_t(hcap - 1);
    v.brick_capacity = uint32_t(blocks);
    hipLaunchKernelGGL(k_fill_u64, grid_for(hcap), dim3(256), 0, stream, v.hash_keys, size_t(hcap), kEmptyKey);
    HIP_TRY(hipMemsetAsync(v.live_bits, 0, hcap / 8, stream));
    HIP_TRY(hipMemsetAsync(v.brick_sum, 0, hcap * 512, stream));
    HIP_TRY(hipMemsetAsync(v.brick_cnt, 0, hcap * 256, stream));
    HIP_TRY(hipMemsetAsync(v.brick_pack, 0, hcap * 512, stream));
    HIP_TRY(hipMemsetAsync(v.brick_count, 0, kCountStripes * 64, stream));
    HIP_TRY(hipGetLastError());
    return CHAD_OK;
}

// A table's bricks released and zeroed (k_clear_bricks), then its flag word: the "exact rows in use" bit always goes, the sticky
// overflow bit only at a reset of the map.  The flag reset is a launch of its own BEHIND the clear (every workgroup of the clear reads
// the exact-rows bit: it may not change under them); where the cleared table becomes the SPARE the reset is left to spare_ready(),
// which runs when the table is swapped in -- off the path of the next frames' updates.
inline void enqueue_clear(const MapView& v, hipStream_t s, uint32_t live = ~0u) { hipLaunchKernelGGL(k_clear_bricks, dim3(walk_grid(v.hash_mask, live)), dim3(256), 0, s, v); }
inline void enqueue_flag_reset(const MapView& v, hipStream_t s, bool overflow_too) {
    hipLaunchKernelGGL(k_table_flags, dim3(1), dim3(1), 0, s, v.overflow, overflow_too ? 0u : uint32_t(kTableOverflowed));
}

// Octree::clear() (octree.hpp:22-29): forget every block of the active submap.  Fully asynchronous.
chad_status clear_active(chad_map* m, bool clear_overflow = false) {
    TRY(exclusive_begin(m));
    enqueue_clear(m->view, m->stream);
    enqueue_flag_reset(m->view, m->stream, clear_overflow);
    HIP_TRY(hipGetLastError());
    m->pending_clear = false; m->pending_clear_overflow = false;   // whatever was pending has just happened
    return exclusive_end(m, m->stream);
}

// A reset not yet carried out (chad_map_reset defers it to the next insert) is carried out now, on the map's stream.
chad_status flush_pending_clear(chad_map* m) {
    if (!m->pending_clear) return CHAD_OK;
    return clear_active(m, m->pending_clear_overflow);
}

chad_status read_block_count(chad_map* m, uint32_t* blocks) {
    uint32_t stripes[kCountStripes * 16];
    TRY(flush_pending_clear(m));
    TRY(join_streams(m));   // updates run on the slots' streams
    HIP_TRY(hipMemcpyAsync(stripes, m->view.brick_count, sizeof(stripes), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    uint32_t v = 0;
    for (uint32_t k = 0; k < kCountStripes; k++) v += stripes[k * 16];
    *blocks = v;
    return CHAD_OK;
}

chad_status finalize_closed(chad_map* m);  // (further down, with the rest of Submap::finalize)
chad_status ensure_aux_sort(chad_map* m, size_t n);

void free_submap(Submap& s) {
    if (!s.borrowed) { (void)chad_free(s.keys); (void)chad_free(s.sum); (void)chad_free(s.cnt); }
    s = Submap{};
}

// A closed submap's dense copy under CHAD_FLAG_DAG_ONLY lives only until it has been folded into the DAG, a moment later: it is
// carved out of a grow-only arena instead of being allocated and freed (hipMalloc / hipFree of hundreds of MB stall the whole
// device, once per roll-over: every eleven frames in BASELINE config 5).
chad_status borrow_submap(chad_map* m, Submap& s, uint32_t blocks) {
    s.blocks = blocks; s.borrowed = true;
    const size_t nb = std::max<uint32_t>(blocks, 1);
    if (Arena::padded(nb * 8) + 2 * Arena::padded(nb * 256) > m->archive_arena.cap) HIP_TRY(hipStreamSynchronize(m->stream));   // (the arena is about to move)
    HIP_TRY(m->archive_arena.reserve(Arena::padded(nb * 8) + 2 * Arena::padded(nb * 256)));
    s.keys = m->archive_arena.take<uint64_t>(nb); s.sum = m->archive_arena.take<float>(nb * 64); s.cnt = m->archive_arena.take<uint32_t>(nb * 64);
    return CHAD_OK;
}

chad_status alloc_submap(Submap& s, uint32_t blocks) {
    s.blocks = blocks;
    const size_t nb = std::max<uint32_t>(blocks, 1);
    HIP_TRY(chad_malloc(&s.keys, nb * 8));
    HIP_TRY(chad_malloc(&s.sum, nb * 64 * 4));
    HIP_TRY(chad_malloc(&s.cnt, nb * 64 * 4));
    return CHAD_OK;
}

// Dense copy of the active submap's bricks (they live scattered over the hash slots), in no particular order.
chad_status compact_active(chad_map* m, Submap& s, uint32_t blocks) {
    TRY(alloc_submap(s, blocks));
    if (!blocks) return CHAD_OK;
    HIP_TRY(hipMemsetAsync(m->d_cursor, 0, 4, m->stream));
    hipLaunchKernelGGL(k_compact_bricks, dim3(walk_grid(m->view.hash_mask)), dim3(256), 0, m->stream, m->view, m->d_cursor, blocks, s.keys, s.sum, s.cnt, m->trunc);
    HIP_TRY(hipGetLastError());
    return CHAD_OK;
}

// CHAD_FLAG_DAG_ONLY: what is closed is folded into the DAG right away and its dense copy released, so that device memory
// holds the active submap only -- what the reference keeps of a closed submap is its DAG too (tsdf.cpp:53-54).
chad_status release_finalized(chad_map* m) {
    if (!(m->flags & CHAD_FLAG_DAG_ONLY)) return CHAD_OK;
    TRY(finalize_closed(m));   // synchronises with the archive copies of everything it folds
    for (Submap& s : m->closed) {
        if (s.released) continue;
        const uint32_t blocks = s.blocks;
        free_submap(s);
        s.blocks = blocks; s.released = true;
    }
    return CHAD_OK;
}

// Second half of a roll-over: the closing table's block count is on the host (or is waited for if `block`): size the
// archive, compact the table into it and clear it -- all on the map's stream, beside the frames that already fill the new
// active table -- and keep the cleared table as the next spare.
chad_status merge_advance(chad_map* m, bool block);   // (below: the multi-GPU form of the same second half)
chad_status fold_advance(chad_map* m, bool block);    // (with Submap::finalize, further down)
chad_status fold_archived(chad_map* m);
chad_status close_into_dag(chad_map* m, uint32_t blocks);
// the map's stream is about to use the spare table (its clear ran on this stream, whichever route closed it)
inline chad_status spare_ready(chad_map* m, bool at_reset = false) {
    // the cleared table's flag word, behind its clear (stream order) and before it takes updates: at a reset everything goes (one
    // memset), at a roll-over the sticky overflow bit stays unless the table was the active one when the map was last reset
    if (at_reset || m->spare_flags == 2) HIP_TRY(hipMemsetAsync(m->spare.overflow, 0, 4, m->stream));
    else if (m->spare_flags == 1) enqueue_flag_reset(m->spare, m->stream, false);
    m->spare_flags = 0;
    return CHAD_OK;
}
chad_status finish_closing(chad_map* m, bool block) {
    if (!m->closing) return CHAD_OK;
    if (m->closing_comm) return merge_advance(m, block);
    if (block) HIP_TRY(hipEventSynchronize(m->closing_counted));
    else if (hipEventQuery(m->closing_counted) != hipSuccess) { (void)hipGetLastError(); return CHAD_OK; }
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < kCountStripes; k++) blocks += m->closing_count[k * 16];
    m->overflow_latched |= m->closing_count[kCountStripes * 16] & kTableOverflowed;   // a submap that lost blocks stays reported after its table has gone back to the pool
    // CHAD_FLAG_DAG_ONLY: no dense copy at all -- cluster words straight from the table, folded into the DAG beside the next frames
    if ((m->flags & CHAD_FLAG_DAG_ONLY) && !m->dag_on_host) return close_into_dag(m, blocks);
    Submap s;
    if (m->flags & CHAD_FLAG_DAG_ONLY) TRY(borrow_submap(m, s, blocks)); else TRY(alloc_submap(s, blocks));
    if (blocks) {
        HIP_TRY(hipMemsetAsync(m->d_cursor, 0, 4, m->stream));
        hipLaunchKernelGGL(k_compact_bricks, dim3(walk_grid(m->closing_view.hash_mask)), dim3(256), 0, m->stream, m->closing_view, m->d_cursor, blocks, s.keys, s.sum, s.cnt, m->trunc);
    }
    enqueue_clear(m->closing_view, m->stream);
    HIP_TRY(hipGetLastError());
    m->closed.push_back(s);
    m->spare = m->closing_view; m->has_spare = true;   // usable behind the clear: whoever swaps it in does so on this stream
    m->spare_flags = std::max(m->spare_flags, 1);
    m->closing_view = MapView{}; m->closing = false;
    return release_finalized(m);
}

inline void staging_restart(chad_map* m);   // (defined with fold_if_due's bookkeeping below)

// Submap roll-over (tsdf.cpp:52-57): archive the active submap's bricks and start an empty one -- by swapping tables, so
// that the frames behind the roll-over do not wait for the archive (a synchronous roll-over drained the three frames in
// flight, sized the archive on the host and compacted: ~1 ms per roll-over, 13 % of the cfg3 stream).
chad_status close_active(chad_map* m) {
    TRY(finish_closing(m, true));          // at most one submap closes at a time (roll-overs are many frames apart)
    TRY(flush_pending_clear(m));
    if (!m->has_spare) {                   // first roll-over, or the table has grown since: an empty table of the same size
        if (alloc_map_storage(m, m->spare, m->brick_capacity, m->stream) != CHAD_OK) {
            // no room for a second table: the synchronous way
            free_map_storage(m, m->spare);
            g_error.clear();
            uint32_t blocks = 0;
            TRY(read_block_count(m, &blocks));
            Submap s;
            TRY(compact_active(m, s, blocks));
            m->closed.push_back(s);
            TRY(clear_active(m));
            return release_finalized(m);
        }
        m->has_spare = true;
    }
    TRY(spare_ready(m));
    TRY(exclusive_begin(m));               // the map's stream waits for every update enqueued so far ...
    HIP_TRY(hipMemcpyAsync(m->closing_count, m->view.brick_count, kCountStripes * 64, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(m->closing_count + kCountStripes * 16, m->view.overflow, 4, hipMemcpyDeviceToHost, m->stream));   // ... and whether it overflowed (final: its updates are done)
    HIP_TRY(hipEventRecord(m->closing_counted, m->stream));
    m->closing_view = m->view; m->closing = true;
    m->view = m->spare; m->spare = MapView{}; m->has_spare = false;
    *m->h_live_blocks = 0;
    staging_restart(m);
    return exclusive_end(m, m->stream);    // ... and later updates (into the new table) wait for this point: the spare is ready
}


// ---- multi-GPU: the exchange at a submap close -------------------------------------------------------
// Collective: every rank of the communicator reaches it at the same point of the frame stream (the roll-over rule depends on the
// poses only, which every rank sees).  Afterwards every rank holds the same closed submap -- the sum of all ranks' partial bricks,
// bit for bit what one GPU would have archived from the same frames.  It runs like the local archive, beside the next frames:
// merge_begin swaps the tables (the closing table is this rank's share), merge_advance enqueues one stage at a time on the map's
// stream, each as soon as the previous stage's numbers have reached the host:
//   stage 0 -> 1  (local block count known)  compact the share raw (64-bit sums); all-gather the ranks' block counts
//   stage 1 -> 2  (all counts known)         all-gather the block keys (lists padded to the longest); sorted union: radix sort + distinct
//   stage 2 -> done (union size known)       scatter the share into dense [U x 64] buffers over the union; all-reduce(sum) of the i64
//                                            sums and u32 counts; convert to the archive's format exactly as the local archive does
// Every rank issues the same four collectives in the same order; between the stages the host only reads a few pinned words.
#define COMM_TRY(expr) do { if (!(expr)) return fail(CHAD_ERR_HIP, "multi-GPU exchange: " + c->error); } while (0)

chad_status merge_begin(chad_map* m, chad_comm* c) {
    if (c->device != m->device) return fail(CHAD_ERR_INVALID_ARG, "the communicator was created on another device");
    TRY(finish_closing(m, true));          // at most one submap closes at a time
    TRY(flush_pending_clear(m));
    if (!m->has_spare) {
        if (alloc_map_storage(m, m->spare, m->brick_capacity, m->stream) != CHAD_OK) {
            free_map_storage(m, m->spare);
            return fail(CHAD_ERR_OUT_OF_MEMORY, "multi-GPU exchange: no room for the spare table the closing submap is swapped against");
        }
        m->has_spare = true;
    }
    TRY(spare_ready(m));
    TRY(exclusive_begin(m));               // the map's stream waits for every update enqueued so far ...
    HIP_TRY(hipMemcpyAsync(m->closing_count, m->view.brick_count, kCountStripes * 64, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(m->closing_count + kCountStripes * 16, m->view.overflow, 4, hipMemcpyDeviceToHost, m->stream));   // ... and whether it overflowed (final: its updates are done)
    HIP_TRY(hipEventRecord(m->closing_counted, m->stream));
    m->closing_view = m->view; m->closing = true; m->closing_comm = c; m->merge_stage = 0;
    // a sharded map that keeps closed submaps as their DAG only exchanges what exists and folds once, on rank 0 (sparse_stages); one that
    // keeps the archives reduces dense bricks so that every rank ends up with the merged submap (merge_stages)
    static const bool dense_always = std::getenv("CHAD_DEBUG_DENSE_EXCHANGE") != nullptr;
    m->closing_sparse = (m->flags & CHAD_FLAG_DAG_ONLY) && !m->dag_on_host && !dense_always;
    m->view = m->spare; m->spare = MapView{}; m->has_spare = false;
    *m->h_live_blocks = 0;
    staging_restart(m);
    return exclusive_end(m, m->stream);    // ... and later updates (into the new table) wait for this point only
}

// A stage failed on THIS rank (out of memory for its scratch, an implausible count, a transport error): the closing state is torn
// down -- the share is dropped, its table cleared and kept as the spare, an empty submap stands in for the lost one so that submap
// indices stay aligned with their poses -- so that this rank reports the error once instead of re-entering the stage on every later
// insert.  The remaining collectives of the exchange are NOT issued: the peers' next collective waits for this rank (RCCL's own
// watchdog, NCCL_TIMEOUT, ends that wait, or their own CHAD_COMM_TIMEOUT_S): ANY failure of a stage on one rank -- a purely local one
// like its scratch not fitting included -- is fatal for the sharded map as a whole (INTEGRATION.md says so).
static void merge_abort(chad_map* m) {
    if (!m->closing) return;
    const bool sparse = m->closing_sparse, not_rank0 = m->closing_comm && m->closing_comm->rank != 0;
    if (chad_comm* c = m->closing_comm) c->abort();   // (RCCL: a collective stuck on the stream is killed with its communicator; the communicator is dead from here on)
    m->closing_sparse = false;
    m->comm_wait_event = nullptr;
    enqueue_clear(m->closing_view, m->stream);
    (void)hipGetLastError();
    m->spare_flags = std::max(m->spare_flags, 1);
    Submap lost;
    if (sparse) {
        // the sparse exchange keeps no bricks of a closed submap: the lost one is a released place holder like every other, and on the
        // ranks that hold no DAG so are its roots -- a dense archive here would later be folded into a stray DAG of this rank's own
        lost.released = true;
        if (not_rank0) { m->dag_roots.push_back({0u, 0u}); m->dag_remote = true; }
    } else if (alloc_submap(lost, 0) != CHAD_OK) { free_submap(lost); (void)hipGetLastError(); }
    m->closed.push_back(lost);
    m->spare = m->closing_view; m->has_spare = true;
    m->closing_view = MapView{}; m->closing = false; m->closing_comm = nullptr; m->merge_stage = 0;
}

// Look at / wait for an event of the exchange.  A wait is bounded (CHAD_COMM_TIMEOUT_S, default 120 s, 0 = for ever): a peer that never
// issues its collective -- it crashed, it took another branch -- must end in an error on this rank, not in a hang (the stage is then
// abandoned: merge_abort, which also tears the RCCL communicator down so that the stuck collective leaves the stream).
static double comm_timeout_s() {
    static const double limit = std::getenv("CHAD_COMM_TIMEOUT_S") ? std::atof(std::getenv("CHAD_COMM_TIMEOUT_S")) : 120.0;
    return limit;
}
// The clock runs from the first look at an event, and a look that does not block applies the limit too: a rank that keeps inserting
// (every insert polls the exchange) notices a silent peer after the same time as one that waits.
static chad_status comm_wait(chad_map* m, hipEvent_t ev, bool block) {
    if (m->comm_wait_event != ev) { m->comm_wait_event = ev; m->comm_wait_since = std::chrono::steady_clock::now(); }
    for (uint32_t polls = 0;; polls++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) { m->comm_wait_event = nullptr; return CHAD_OK; }
        (void)hipGetLastError();
        if (e != hipErrorNotReady) HIP_TRY(e);
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - m->comm_wait_since).count();
        if (comm_timeout_s() > 0 && waited > comm_timeout_s()) {
            m->comm_wait_event = nullptr;
            return fail(CHAD_ERR_HIP, "a stage of the exchange did not complete within " + std::to_string(int(comm_timeout_s())) + " s (a peer that never issued its collective?)");
        }
        if (!block) return CHAD_OK;
        if (polls > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

static chad_status merge_stages(chad_map* m, bool block);
static chad_status sparse_stages(chad_map* m, bool block);
chad_status merge_advance(chad_map* m, bool block) {
    const chad_status s = m->closing_sparse ? sparse_stages(m, block) : merge_stages(m, block);
    if (s != CHAD_OK) { const std::string why = g_error; merge_abort(m); g_error = "multi-GPU exchange abandoned on this rank: " + why; }
    return s;
}

static chad_status merge_stages(chad_map* m, bool block) {
    chad_comm* c = m->closing_comm;
    hipStream_t s = m->stream;
    const uint32_t world = uint32_t(c->world);
    for (;;) {
        hipEvent_t pending = m->merge_stage == 0 ? m->closing_counted : m->merge_event;
        TRY(comm_wait(m, pending, block));
        if (hipEventQuery(pending) != hipSuccess) { (void)hipGetLastError(); return CHAD_OK; }
        if (m->merge_stage == 0) {
            uint32_t nb = 0;
            for (uint32_t k = 0; k < kCountStripes; k++) nb += m->closing_count[k * 16];
            m->overflow_latched |= m->closing_count[kCountStripes * 16] & kTableOverflowed;
            m->merge_blocks = nb;
            HIP_TRY(m->merge_a.reserve(Arena::padded(size_t(nb) * 8) + Arena::padded(size_t(nb) * 512) + Arena::padded(size_t(nb) * 256) + Arena::padded(8) + Arena::padded(8 * size_t(world))));
            m->merge_keys = m->merge_a.take<uint64_t>(nb); m->merge_sum = m->merge_a.take<long long>(size_t(nb) * 64); m->merge_cnt = m->merge_a.take<uint32_t>(size_t(nb) * 64);
            uint64_t* d_n = m->merge_a.take<uint64_t>(1);
            uint64_t* d_all_n = m->merge_a.take<uint64_t>(world);
            if (nb) {
                HIP_TRY(hipMemsetAsync(m->d_cursor, 0, 4, s));
                hipLaunchKernelGGL(k_compact_bricks_raw, dim3(walk_grid(m->closing_view.hash_mask)), dim3(256), 0, s, m->closing_view, m->d_cursor, nb, m->merge_keys, m->merge_sum, m->merge_cnt);
                HIP_TRY(hipGetLastError());
            }
            m->merge_host[17] = nb;
            HIP_TRY(hipMemcpyAsync(d_n, &m->merge_host[17], 8, hipMemcpyHostToDevice, s));
            COMM_TRY(c->all_gather(d_n, d_all_n, 8, s));
            HIP_TRY(hipMemcpyAsync(m->merge_host, d_all_n, 8 * size_t(world), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(m->merge_event, s));
            m->merge_stage = 1;
        } else if (m->merge_stage == 1) {
            uint64_t total = 0, longest = 0;
            RankCounts rc{};
            for (uint32_t r = 0; r < world; r++) {
                const uint64_t n = m->merge_host[r];
                if (n >= (1ull << 31)) return fail(CHAD_ERR_HIP, "multi-GPU exchange: implausible block count from a rank");
                rc.count[r] = uint32_t(n); rc.begin[r] = uint32_t(total);
                total += n; longest = std::max(longest, n);
            }
            if (total >= (1ull << 30)) return fail(CHAD_ERR_CAPACITY, "multi-GPU exchange: more than 2^30 blocks over all ranks");
            m->merge_total = total;
            const uint32_t chunks = uint32_t((total + kUniqueChunk - 1) / kUniqueChunk);
            HIP_TRY(m->merge_b.reserve(Arena::padded(longest * 8) + Arena::padded(longest * 8 * world) + Arena::padded(size_t(chunks) * 4) + Arena::padded(4) + Arena::padded(total * 8)));
            uint64_t* send = m->merge_b.take<uint64_t>(longest);
            uint64_t* recv = m->merge_b.take<uint64_t>(longest * world);
            m->merge_heads = m->merge_b.take<uint32_t>(chunks);
            uint32_t* d_union_n = m->merge_b.take<uint32_t>(1);
            m->merge_sorted = m->merge_b.take<uint64_t>(total);
            if (total) {
                // all ranks' keys (lists padded to the longest), then the sorted union -- the same on every rank
                hipLaunchKernelGGL(k_fill_u64, grid_for(longest), dim3(256), 0, s, send, size_t(longest), kEmptyKey);
                if (m->merge_blocks) HIP_TRY(hipMemcpyAsync(send, m->merge_keys, size_t(m->merge_blocks) * 8, hipMemcpyDeviceToDevice, s));
                COMM_TRY(c->all_gather(send, recv, longest * 8, s));
                m->pt.exchange_bytes += longest * 8;
                TRY(ensure_aux_sort(m, total));
                hipLaunchKernelGGL(k_concat_keys, grid_for(longest), dim3(256), 0, s, recv, uint32_t(longest), rc, world, m->aux_sort.keys[0]);
                RadixSortSignals sig;
                sig.fault = m->d_overflow;
                radix_sort_enqueue(m->aux_sort, uint32_t(total), kMaxPasses, nullptr, s, 0, 0, sig);   // an even number of passes: sorted keys end up in keys[0]
                HIP_TRY(hipMemcpyAsync(m->merge_sorted, m->aux_sort.keys[0], size_t(total) * 8, hipMemcpyDeviceToDevice, s));   // aux_sort is free again behind this copy
                hipLaunchKernelGGL(k_unique_count, dim3(std::min(chunks, 2048u)), dim3(256), 0, s, m->merge_sorted, uint32_t(total), m->merge_heads);
                hipLaunchKernelGGL(k_unique_offsets, dim3(1), dim3(256), 0, s, m->merge_heads, chunks, d_union_n);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(&m->merge_host[16], d_union_n, 4, hipMemcpyDeviceToHost, s));
            } else m->merge_host[16] = 0;   // (nothing to gather: every rank sees total == 0 and skips the collective alike)
            HIP_TRY(hipEventRecord(m->merge_event, s));
            m->merge_stage = 2;
        } else {
            const uint32_t n_union = m->merge_total ? uint32_t(m->merge_host[16] & 0xffffffffull) : 0u;
            Submap arch;
            TRY(alloc_submap(arch, n_union));
            if (n_union) {
                const uint32_t chunks = uint32_t((m->merge_total + kUniqueChunk - 1) / kUniqueChunk);
                const size_t cells = size_t(n_union) * 64;
                HIP_TRY(m->merge_c.reserve(Arena::padded(cells * 8) + Arena::padded(cells * 4)));
                long long* dense_sum = m->merge_c.take<long long>(cells);
                uint32_t* dense_cnt = m->merge_c.take<uint32_t>(cells);
                hipLaunchKernelGGL(k_unique_write, dim3(std::min(chunks, 2048u)), dim3(256), 0, s, m->merge_sorted, uint32_t(m->merge_total), m->merge_heads, arch.keys);
                // this rank's bricks into dense buffers over the union; the all-reduce: 64-bit integer sums and counts, exact in any order
                HIP_TRY(hipMemsetAsync(dense_sum, 0, cells * 8, s));
                HIP_TRY(hipMemsetAsync(dense_cnt, 0, cells * 4, s));
                if (m->merge_blocks) hipLaunchKernelGGL(k_scatter_bricks_dense, grid_for(size_t(m->merge_blocks) * 64), dim3(256), 0, s, m->merge_keys, m->merge_sum, m->merge_cnt, m->merge_blocks,
                                                        arch.keys, n_union, dense_sum, dense_cnt);
                HIP_TRY(hipGetLastError());
                COMM_TRY(c->all_reduce(dense_sum, cells, CHAD_REDUCE_I64, s));
                COMM_TRY(c->all_reduce(dense_cnt, cells, CHAD_REDUCE_U32, s));
                m->pt.exchange_bytes += cells * 12; m->pt.exchange_union_blocks = n_union;
                // the archive's format, converted exactly as the single-GPU archive converts (k_compact_bricks)
                hipLaunchKernelGGL(k_dense_to_archive, grid_for(cells), dim3(256), 0, s, dense_sum, dense_cnt, cells, m->trunc, arch.sum, arch.cnt);
            }
            enqueue_clear(m->closing_view, s);
            HIP_TRY(hipGetLastError());
            m->spare_flags = std::max(m->spare_flags, 1);
            m->closed.push_back(arch);
            m->spare = m->closing_view; m->has_spare = true;   // usable behind the clear: whoever swaps it in does so on this stream
            m->closing_view = MapView{}; m->closing = false; m->closing_comm = nullptr; m->merge_stage = 0;
            m->pt.exchanges++;
            return release_finalized(m);
        }
    }
}
#undef COMM_TRY


// ---- the sparse exchange (exchange_kernels.hpp): frame-sharded maps that keep closed submaps as their DAG only -----------------------
// Stages, each enqueued on the map's stream when the previous one's numbers have reached the host (polled by every insert, waited
// for by whatever needs the closed submap):
//   0 -> 1  (local block count known)  the closing table's keys in Morton order; all-gather of every rank's count + 64 sample keys
//   1 -> 2  (all heads known)          splitters (the same on every rank); this rank's blocks per range; their voxel masks and the
//                                      payload's layout; all-gather of the G x G matrix "blocks and words from rank a for rank b"
//   2 -> 3  (matrix known)             the slices packed; point-to-point exchange; sorted union of the keys this rank received
//   3 -> 4  (union size known)         the slices added into dense accumulators over the union; cluster words of the range;
//                                      all-gather of the ranges' sizes
//   4 -> done (sizes known)            the ranges' cluster words to rank 0, which concatenates them -- Morton order -- and folds the
//                                      submap into its DAG (beside the next frames, like a single GPU's roll-over); the closing
//                                      table is cleared and becomes the spare.  The other ranks hold no DAG (dag_remote).
#define COMM_TRY(expr) do { if (!(expr)) return fail(CHAD_ERR_HIP, "multi-GPU exchange: " + c->error); } while (0)
static chad_status sparse_stages(chad_map* m, bool block) {
    chad_comm* c = m->closing_comm;
    hipStream_t s = m->stream;
    const uint32_t world = uint32_t(c->world), me = uint32_t(c->rank);
    chad_map::SparseExchange& x = m->sx;
    uint64_t* H = m->merge_host;
    constexpr uint32_t kHead = kXHeadWords, kHSplit = 1024, kHMatrix = 1056, kHAllU = 1600, kHUnion = 1640;
    const bool saturate = (m->flags & CHAD_FLAG_DAG_SATURATE_WEIGHTS) != 0;
    for (;;) {
        hipEvent_t pending = m->merge_stage == 0 ? m->closing_counted : m->merge_event;
        TRY(comm_wait(m, pending, block));
        if (hipEventQuery(pending) != hipSuccess) { (void)hipGetLastError(); return CHAD_OK; }
        // device side of the small words (stage 0 makes them): head | all heads | splitters | row | matrix | my union size | all union sizes
        uint64_t* d_head = x.d_small; uint64_t* d_all_heads = d_head + kHead; uint64_t* d_split = d_all_heads + kHead * 16; uint64_t* d_row = d_split + 16;
        uint64_t* d_matrix = d_row + 32; uint64_t* d_u = d_matrix + 512; uint64_t* d_all_u = d_u + 2;
        if (m->merge_stage == 0) {
            uint32_t nb = 0;
            for (uint32_t k = 0; k < kCountStripes; k++) nb += m->closing_count[k * 16];
            const uint32_t flags = m->closing_count[kCountStripes * 16];
            m->overflow_latched |= flags & kTableOverflowed;
            x = chad_map::SparseExchange{};
            x.nb = nb; x.wide = (flags & kTableExactRows) ? 1u : 0u;
            const uint32_t chunks = (nb + kUniqueChunk - 1) / kUniqueChunk;
            const size_t nbp = size_t(nb) + 1;
            HIP_TRY(m->merge_a.reserve(Arena::padded(nbp * 8) + 3 * Arena::padded(nbp * 4) + Arena::padded(nbp * 8) + Arena::padded(17 * 4) + Arena::padded((size_t(chunks) + 1) * 4) + Arena::padded(4) + Arena::padded(2048 * 8)));
            x.keys_sorted = m->merge_a.take<uint64_t>(nbp);
            uint32_t* slots = m->merge_a.take<uint32_t>(nbp);
            x.slot_sorted = m->merge_a.take<uint32_t>(nbp);
            x.prefix = m->merge_a.take<uint32_t>(nbp);
            x.mask = m->merge_a.take<uint64_t>(nbp);
            x.first_block = m->merge_a.take<uint32_t>(17);
            x.chunk_sums = m->merge_a.take<uint32_t>(size_t(chunks) + 1);
            x.d_words = m->merge_a.take<uint32_t>(1);
            x.d_small = m->merge_a.take<uint64_t>(2048);
            d_head = x.d_small; d_all_heads = d_head + kHead;
            if (nb) {   // this rank's live slots in key order (close_into_dag's step A, first half)
                m->in_fold_setup = true;   // (nothing sorts on another stream in this mode: no need to wait for a fold in flight)
                const chad_status sorter = ensure_aux_sort(m, nb);
                m->in_fold_setup = false;
                TRY(sorter);
                RadixSortSignals sig;
                sig.fault = m->d_overflow;
                sig.dep = aux_sort_dep(m->aux_sort);
                HIP_TRY(hipMemsetAsync(m->d_cursor + 1, 0, 4, s));
                hipLaunchKernelGGL(k_table_keys, dim3(std::max(1u, std::min(2048u, ((m->closing_view.hash_mask >> 5) + 256u) / 256u))), dim3(256), 0, s, m->closing_view, m->d_cursor + 1, nb, m->aux_sort.keys[0], slots);
                radix_sort_enqueue(m->aux_sort, nb, kMaxPasses, nullptr, s, 0, 0, sig);
                HIP_TRY(hipMemcpyAsync(x.keys_sorted, m->aux_sort.keys[0], size_t(nb) * 8, hipMemcpyDeviceToDevice, s));
                hipLaunchKernelGGL(k_xchg_order, grid_for(nb), dim3(256), 0, s, (const uint32_t*)m->aux_sort.idx[0], (const uint32_t*)slots, nb, x.slot_sorted);   // (the sorter is free again behind this)
            }
            hipLaunchKernelGGL(k_xchg_head, dim3(1), dim3(128), 0, s, (const uint64_t*)x.keys_sorted, nb, x.wide, d_head);
            HIP_TRY(hipGetLastError());
            COMM_TRY(c->all_gather(d_head, d_all_heads, kHead * 8, s));
            m->pt.exchange_bytes += kHead * 8;
            HIP_TRY(hipMemcpyAsync(H, d_all_heads, size_t(kHead) * 8 * world, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(m->merge_event, s));
            m->merge_stage = 1;
        } else if (m->merge_stage == 1) {
            // the splitters: plain deterministic host arithmetic on words every rank holds alike (exchange_plan.hpp)
            if (!xchg_splitters(H, world, &H[kHSplit])) return fail(CHAD_ERR_HIP, "multi-GPU exchange: implausible block count from a rank");
            for (uint32_t r = 0; r < world; r++) x.wide_of[r] = uint32_t(H[size_t(r) * kHead] >> 32) & 1u;
            if (world > 1) HIP_TRY(hipMemcpyAsync(d_split, &H[kHSplit], size_t(world - 1) * 8, hipMemcpyHostToDevice, s));
            const uint32_t nb = x.nb, chunks = (nb + kUniqueChunk - 1) / kUniqueChunk;
            hipLaunchKernelGGL(k_xchg_ranges, dim3(1), dim3(64), 0, s, (const uint64_t*)x.keys_sorted, nb, (const uint64_t*)d_split, world, x.first_block);
            if (nb) {
                hipLaunchKernelGGL(k_xchg_masks, grid_for(size_t(nb) * 16, 256, 4096), dim3(256), 0, s, m->closing_view, (const uint32_t*)x.slot_sorted, nb, x.mask);
                hipLaunchKernelGGL(k_xchg_chunk_sums, dim3(std::min(chunks, 2048u)), dim3(256), 0, s, (const uint64_t*)x.mask, nb, x.chunk_sums);
                hipLaunchKernelGGL(k_unique_offsets, dim3(1), dim3(256), 0, s, x.chunk_sums, chunks, x.d_words);
            }
            hipLaunchKernelGGL(k_xchg_prefix, dim3(std::max(1u, std::min(chunks, 2048u))), dim3(256), 0, s, (const uint64_t*)x.mask, nb, (const uint32_t*)x.chunk_sums, (const uint32_t*)x.d_words, x.prefix);
            hipLaunchKernelGGL(k_xchg_row, dim3(1), dim3(64), 0, s, (const uint32_t*)x.first_block, (const uint32_t*)x.prefix, world, d_row);
            HIP_TRY(hipGetLastError());
            COMM_TRY(c->all_gather(d_row, d_matrix, size_t(world) * 16, s));
            m->pt.exchange_bytes += size_t(world) * 16;
            HIP_TRY(hipMemcpyAsync(&H[kHMatrix], d_matrix, size_t(world) * world * 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(m->merge_event, s));
            m->merge_stage = 2;
        } else if (m->merge_stage == 2) {
            auto blocks_of = [&](uint32_t from, uint32_t to) { return H[kHMatrix + (size_t(from) * world + to) * 2]; };
            auto words_of = [&](uint32_t from, uint32_t to) { return H[kHMatrix + (size_t(from) * world + to) * 2 + 1]; };
            ExchangePlan plan;
            if (!xchg_slice_plan(&H[kHMatrix], x.wide_of, world, me, plan)) return fail(CHAD_ERR_HIP, "multi-GPU exchange: implausible slice size from a rank");
            XchgSlices slices{};
            const size_t send_total = plan.send_total(), recv_total = plan.recv_total();
            uint64_t recv_blocks = 0, mine_check = 0;
            for (uint32_t r = 0; r < world; r++) {
                slices.off[r] = plan.send_off[r]; x.recv_off[r] = plan.recv_off[r];
                x.blocks_from[r] = blocks_of(r, me); x.words_from[r] = words_of(r, me); recv_blocks += x.blocks_from[r];
                mine_check += blocks_of(me, r);
            }
            if (mine_check != x.nb) return fail(CHAD_ERR_HIP, "multi-GPU exchange: this rank's ranges do not add up to its blocks");
            if (recv_blocks >= (1ull << 30)) return fail(CHAD_ERR_CAPACITY, "multi-GPU exchange: more than 2^30 blocks in one rank's range");
            x.recv_blocks = uint32_t(recv_blocks);
            const uint32_t chunks = uint32_t((recv_blocks + kUniqueChunk - 1) / kUniqueChunk);
            HIP_TRY(m->merge_b.reserve(Arena::padded(send_total) + Arena::padded(recv_total) + Arena::padded((recv_blocks + 1) * 8) + Arena::padded((size_t(chunks) + 1) * 4) + Arena::padded(4)));
            x.send = m->merge_b.take<char>(send_total + 8);
            x.recv = m->merge_b.take<char>(recv_total + 8);
            x.sorted = m->merge_b.take<uint64_t>(recv_blocks + 1);
            x.heads = m->merge_b.take<uint32_t>(size_t(chunks) + 1);
            x.d_union_n = m->merge_b.take<uint32_t>(1);
            if (x.nb) hipLaunchKernelGGL(k_xchg_pack, grid_for(size_t(x.nb) * 64, 256, 4096), dim3(256), 0, s, m->closing_view, (const uint32_t*)x.slot_sorted, (const uint64_t*)x.keys_sorted, (const uint64_t*)x.mask,
                                         (const uint32_t*)x.prefix, (const uint32_t*)x.first_block, x.nb, world, x.wide, x.send, slices);
            HIP_TRY(hipGetLastError());
            COMM_TRY(c->exchange_slices(x.send, x.recv, plan, s));
            m->pt.exchange_bytes += plan.bytes_out(int(me), int(world));
            if (recv_blocks) {   // sorted union of the keys this rank received (its own slice among them)
                m->in_fold_setup = true;
                const chad_status sorter = ensure_aux_sort(m, recv_blocks);
                m->in_fold_setup = false;
                TRY(sorter);
                size_t at = 0;
                for (uint32_t r = 0; r < world; r++) {
                    if (x.blocks_from[r]) HIP_TRY(hipMemcpyAsync(m->aux_sort.keys[0] + at, x.recv + x.recv_off[r], size_t(x.blocks_from[r]) * 8, hipMemcpyDeviceToDevice, s));
                    at += x.blocks_from[r];
                }
                RadixSortSignals sig;
                sig.fault = m->d_overflow;
                radix_sort_enqueue(m->aux_sort, uint32_t(recv_blocks), kMaxPasses, nullptr, s, 0, 0, sig);
                HIP_TRY(hipMemcpyAsync(x.sorted, m->aux_sort.keys[0], size_t(recv_blocks) * 8, hipMemcpyDeviceToDevice, s));
                hipLaunchKernelGGL(k_unique_count, dim3(std::min(chunks, 2048u)), dim3(256), 0, s, (const uint64_t*)x.sorted, uint32_t(recv_blocks), x.heads);
                hipLaunchKernelGGL(k_unique_offsets, dim3(1), dim3(256), 0, s, x.heads, chunks, x.d_union_n);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(&H[kHUnion], x.d_union_n, 4, hipMemcpyDeviceToHost, s));
            }
            HIP_TRY(hipEventRecord(m->merge_event, s));
            m->merge_stage = 3;
        } else if (m->merge_stage == 3) {
            const uint32_t n_union = x.recv_blocks ? uint32_t(H[kHUnion] & 0xffffffffull) : 0u;
            if (n_union > x.recv_blocks) return fail(CHAD_ERR_HIP, "multi-GPU exchange: implausible union size");
            x.n_union = n_union;
            const size_t cells = size_t(n_union) * 64, nu = size_t(n_union) + 1;
            const uint32_t uchunks = (n_union + kUniqueChunk - 1) / kUniqueChunk;
            x.out_bytes = xchg_range_bytes(n_union, size_t(n_union) * 8, saturate);   // (room for the worst case; what travels is sized by the words that exist)
            HIP_TRY(m->merge_c.reserve(Arena::padded(x.out_bytes + 8) + Arena::padded(cells * 8 + 8) + Arena::padded(cells * 4 + 8) + (saturate ? 2 : 1) * Arena::padded(nu * 64) + Arena::padded(nu) +
                                       Arena::padded(nu * 8) + Arena::padded(nu * 4) + Arena::padded((size_t(uchunks) + 1) * 4) + Arena::padded(4)));
            x.out = m->merge_c.take<char>(x.out_bytes + 8);
            long long* acc_sum = m->merge_c.take<long long>(cells + 1);
            uint32_t* acc_cnt = m->merge_c.take<uint32_t>(cells + 1);
            uint64_t* rows_tsdf = m->merge_c.take<uint64_t>(nu * 8);
            uint64_t* rows_weight = saturate ? m->merge_c.take<uint64_t>(nu * 8) : nullptr;
            uint8_t* rows_mask = m->merge_c.take<uint8_t>(nu);
            uint64_t* mask64 = m->merge_c.take<uint64_t>(nu);
            uint32_t* first = m->merge_c.take<uint32_t>(nu);
            uint32_t* usums = m->merge_c.take<uint32_t>(size_t(uchunks) + 1);
            uint32_t* d_total = m->merge_c.take<uint32_t>(1);
            uint64_t* out_keys = reinterpret_cast<uint64_t*>(x.out);
            if (n_union) {
                const uint32_t chunks = (x.recv_blocks + kUniqueChunk - 1) / kUniqueChunk;
                hipLaunchKernelGGL(k_unique_write, dim3(std::min(chunks, 2048u)), dim3(256), 0, s, (const uint64_t*)x.sorted, x.recv_blocks, (const uint32_t*)x.heads, out_keys);
                HIP_TRY(hipMemsetAsync(acc_sum, 0, cells * 8, s));
                HIP_TRY(hipMemsetAsync(acc_cnt, 0, cells * 4, s));
                for (uint32_t r = 0; r < world; r++)   // one source after the other: plain adds, exact in any order
                    if (x.blocks_from[r]) hipLaunchKernelGGL(k_xchg_accumulate, grid_for(size_t(x.blocks_from[r]) * 64, 256, 4096), dim3(256), 0, s, (const char*)(x.recv + x.recv_off[r]), uint32_t(x.blocks_from[r]),
                                                             x.wide_of[r], (const uint64_t*)out_keys, n_union, acc_sum, acc_cnt);
                hipLaunchKernelGGL(k_xchg_clusters, grid_for(cells, 256, 4096), dim3(256), 0, s, (const long long*)acc_sum, (const uint32_t*)acc_cnt, n_union, m->trunc, 1.0f / m->trunc /* submap.hpp:23 */,
                                   rows_tsdf, rows_weight, rows_mask);
                // only the octets that hold leaves travel: their places by a scan of the masks' bit counts
                hipLaunchKernelGGL(k_xchg_widen, grid_for(n_union), dim3(256), 0, s, (const uint8_t*)rows_mask, n_union, mask64);
                hipLaunchKernelGGL(k_xchg_chunk_sums, dim3(std::min(uchunks, 2048u)), dim3(256), 0, s, (const uint64_t*)mask64, n_union, usums);
                hipLaunchKernelGGL(k_unique_offsets, dim3(1), dim3(256), 0, s, usums, uchunks, d_total);
            }
            hipLaunchKernelGGL(k_xchg_prefix, dim3(std::max(1u, std::min(uchunks, 2048u))), dim3(256), 0, s, (const uint64_t*)mask64, n_union, (const uint32_t*)usums, (const uint32_t*)d_total, first);
            hipLaunchKernelGGL(k_xchg_compact_rows, grid_for(std::max<size_t>(cells / 8, 1), 256, 4096), dim3(256), 0, s, (const uint64_t*)rows_tsdf, (const uint64_t*)rows_weight, (const uint8_t*)rows_mask,
                               (const uint32_t*)first, n_union, x.out, d_u);
            HIP_TRY(hipGetLastError());
            COMM_TRY(c->all_gather(d_u, d_all_u, 16, s));
            m->pt.exchange_bytes += 16;
            HIP_TRY(hipMemcpyAsync(&H[kHAllU], d_all_u, size_t(world) * 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(m->merge_event, s));
            m->merge_stage = 4;
        } else {
            // the ranges to rank 0, which holds the DAG
            ExchangePlan plan;
            uint64_t total_blocks = 0;
            if (!xchg_gather_plan(&H[kHAllU], world, me, saturate, plan, &total_blocks)) return fail(CHAD_ERR_HIP, "multi-GPU exchange: implausible range size from a rank");
            if (total_blocks >= (1ull << 30)) return fail(CHAD_ERR_CAPACITY, "multi-GPU exchange: more than 2^30 blocks in one submap");
            if (H[kHAllU + 2 * me] != x.n_union) return fail(CHAD_ERR_HIP, "multi-GPU exchange: this rank's range size came back changed");
            const size_t my_bytes = plan.send_bytes[0], recv_total = plan.recv_total();
            if (me == 0) {
                while (m->pending.set) TRY(fold_advance(m, true));   // (at most one submap waits while another is folded: close_into_dag's rule)
                const size_t nb = std::max<uint64_t>(total_blocks, 1), words = nb * 8;
                const size_t need = Arena::padded(nb * 8) + (saturate ? 2 : 1) * Arena::padded(words * 8) + Arena::padded(nb);
                const uint32_t parity = m->closings & 1u;
                Arena& arena = m->cluster_arenas[parity];
                if (need > arena.cap) { HIP_TRY(hipStreamSynchronize(s)); if (m->fold_stream) HIP_TRY(hipStreamSynchronize(m->fold_stream)); }
                HIP_TRY(arena.reserve(need));
                uint64_t* keys = arena.take<uint64_t>(nb);
                uint64_t* tsdf = arena.take<uint64_t>(words);
                uint64_t* weight = saturate ? arena.take<uint64_t>(words) : nullptr;
                uint8_t* mask = arena.take<uint8_t>(nb);
                HIP_TRY(m->merge_b.reserve(Arena::padded(recv_total + 8)));   // (the slices of stage 2 are dead: stage 3's kernels have run)
                char* staging = m->merge_b.take<char>(recv_total + 8);
                if (m->close_t0[parity]) HIP_TRY(hipEventRecord(m->close_t0[parity], s));
                COMM_TRY(c->exchange_slices(x.out, staging, plan, s));
                size_t at = 0;
                for (uint32_t r = 0; r < world; r++) {   // ranges in rank order = ascending Morton order
                    const uint64_t u = H[kHAllU + 2 * r], w = H[kHAllU + 2 * r + 1];
                    if (!u) continue;
                    const char* from = staging + plan.recv_off[r];
                    HIP_TRY(hipMemcpyAsync(keys + at, from, u * 8, hipMemcpyDeviceToDevice, s));
                    hipLaunchKernelGGL(k_xchg_expand_rows, grid_for(u * 8, 256, 4096), dim3(256), 0, s, from, uint32_t(u), uint32_t(w), saturate ? 1u : 0u, tsdf + at * 8, saturate ? weight + at * 8 : nullptr, mask + at);
                    at += u;
                }
                HIP_TRY(hipGetLastError());
                if (m->close_t1[parity]) HIP_TRY(hipEventRecord(m->close_t1[parity], s));
                HIP_TRY(hipEventRecord(m->clusters_ready, s));
                m->pending.keys = keys; m->pending.tsdf = tsdf; m->pending.weight = weight; m->pending.mask = mask; m->pending.blocks = uint32_t(total_blocks); m->pending.parity = parity;
                m->pending.set = !m->dag_broken;
                m->closings++;
            } else {
                COMM_TRY(c->exchange_slices(x.out, nullptr, plan, s));
                m->pt.exchange_bytes += my_bytes;
                m->dag_roots.push_back({0u, 0u});   // (a place holder: the submap's roots are rank 0's to know)
                m->dag_remote = true;
            }
            enqueue_clear(m->closing_view, s);
            HIP_TRY(hipGetLastError());
            m->spare_flags = std::max(m->spare_flags, 1);
            m->spare = m->closing_view; m->has_spare = true;
            m->closing_view = MapView{}; m->closing = false; m->closing_comm = nullptr; m->closing_sparse = false; m->merge_stage = 0;
            Submap gone;   // a closed submap without bricks: it is its DAG (on rank 0)
            gone.blocks = uint32_t(total_blocks); gone.released = true;
            m->closed.push_back(gone);
            m->pt.exchanges++; m->pt.exchange_union_blocks = total_blocks; m->pt.rollovers++;
            return me == 0 ? fold_advance(m, false) : CHAD_OK;
        }
    }
}
#undef COMM_TRY

// The exchange, start to finish (explicit calls; the roll-over inside insert() only begins it).
chad_status merge_and_close(chad_map* m, chad_comm* c) {
    if (!c || !c->exchanges()) return close_active(m);
    TRY(merge_begin(m, c));
    return CHAD_OK;
}

chad_status select_submap(chad_map* m, int32_t submap, uint32_t* blocks, const uint64_t** keys, const float** sum, const uint32_t** cnt) {
    if (submap < 0) {  // the active submap: a dense snapshot, kept until the next call
        TRY(read_block_count(m, blocks));
        free_submap(m->active_copy);
        TRY(compact_active(m, m->active_copy, *blocks));
        HIP_TRY(hipStreamSynchronize(m->stream));
        *keys = m->active_copy.keys; *sum = m->active_copy.sum; *cnt = m->active_copy.cnt;
        return CHAD_OK;
    }
    TRY(finish_closing(m, true));
    if (size_t(submap) >= m->closed.size()) return fail(CHAD_ERR_INVALID_ARG, "submap index out of range");
    HIP_TRY(hipStreamSynchronize(m->stream));
    const Submap& s = m->closed[size_t(submap)];
    if (s.released) return fail(CHAD_ERR_INVALID_ARG, "the submap's bricks were released (CHAD_FLAG_DAG_ONLY): read it back with chad_map_dag_leaves");
    *blocks = s.blocks; *keys = s.keys; *sum = s.sum; *cnt = s.cnt;
    return CHAD_OK;
}

struct HostBricks {
    std::vector<uint64_t> keys; std::vector<float> sum; std::vector<uint32_t> cnt; std::vector<uint32_t> order;
};

chad_status fetch_bricks(chad_map* m, int32_t submap, HostBricks& hb) {
    uint32_t blocks = 0;
    const uint64_t* keys; const float* sum; const uint32_t* cnt;
    TRY(select_submap(m, submap, &blocks, &keys, &sum, &cnt));
    hb.keys.resize(blocks); hb.sum.resize(size_t(blocks) * 64); hb.cnt.resize(size_t(blocks) * 64);
    if (blocks) {
        HIP_TRY(hipMemcpy(hb.keys.data(), keys, size_t(blocks) * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hb.sum.data(), sum, size_t(blocks) * 256, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hb.cnt.data(), cnt, size_t(blocks) * 256, hipMemcpyDeviceToHost));
    }
    hb.order.resize(blocks);
    std::iota(hb.order.begin(), hb.order.end(), 0u);
    std::sort(hb.order.begin(), hb.order.end(), [&](uint32_t a, uint32_t b) { return hb.keys[a] < hb.keys[b]; });
    return CHAD_OK;
}

inline bool times_stages(const chad_map* m) { return (m->flags & (CHAD_FLAG_TIMING | CHAD_FLAG_VERBOSE)) != 0; }
inline bool times_kernels(const chad_map* m) { return (m->flags & CHAD_FLAG_TIME_KERNELS) != 0; }
// kernel `id` of the insert being enqueued: its event pair (null when kernels are not timed)
inline hipEvent_t* kernel_events(chad_map* m, int id) {
    if (!times_kernels(m)) return nullptr;
    const int at = int(m->ktimed_end % chad_map::kKernelRing);
    m->kev_launched[at] |= 1u << id;
    return &m->kev[at][2 * id];
}
// sums of the kernel durations of the inserts not read yet (the streams must be idle)
void collect_kernel_times(chad_map* m) {
    // CHAD_DEBUG_TIMELINE: every timed kernel's begin and end, in microseconds after the first timed frame's first kernel (stderr;
    // scripts/timeline.py turns it into the frames-in-flight picture: concurrency, gaps on each frame's stream)
    static const bool timeline = std::getenv("CHAD_DEBUG_TIMELINE") != nullptr;
    for (uint64_t f = m->ktimed_begin; f < m->ktimed_end; f++) {
        const int at = int(f % chad_map::kKernelRing);
        if (timeline) {
            static hipEvent_t base = nullptr;
            if (!base) base = m->kev[at][2 * CHAD_KERNEL_BOUNDS];
            for (int id = 0; id < CHAD_KERNEL_COUNT; id++) {
                if (!((m->kev_launched[at] >> id) & 1u)) continue;
                float b = 0, e = 0;
                if (hipEventElapsedTime(&b, base, m->kev[at][2 * id]) == hipSuccess && hipEventElapsedTime(&e, base, m->kev[at][2 * id + 1]) == hipSuccess)
                    std::fprintf(stderr, "timeline frame %llu kernel %d begin %.2f end %.2f\n", (unsigned long long)f, id, b * 1e3, e * 1e3);
                else (void)hipGetLastError();
            }
        }
        for (int id = 0; id < CHAD_KERNEL_COUNT; id++) {
            if (!((m->kev_launched[at] >> id) & 1u)) continue;
            float t = 0;
            if (hipEventElapsedTime(&t, m->kev[at][2 * id], m->kev[at][2 * id + 1]) == hipSuccess) { m->kernel_ms[id] += t; m->kernel_launches[id]++; }
            else (void)hipGetLastError();
        }
        m->kev_launched[at] = 0;
    }
    m->ktimed_begin = m->ktimed_end;
}
// CHAD_DEBUG_HOST_PROFILE: where the host's time per insert goes (microseconds per call and section, printed when a map is destroyed)
struct HostProfile {
    bool on = std::getenv("CHAD_DEBUG_HOST_PROFILE") != nullptr;
    double us[12] = {};
    uint64_t calls = 0;
    std::chrono::steady_clock::time_point last;
    void begin() { if (on) { last = std::chrono::steady_clock::now(); calls++; } }
    void mark(int section) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        us[section] += std::chrono::duration<double, std::micro>(now - last).count();
        last = now;
    }
    void report() const {
        if (!on || !calls) return;
        static const char* const names[12] = {"pose/closing/growth", "bounds+keys", "sort", "gather+normals", "waits/clear", "update", "tail", "reset", "prepare_slot", "", "", ""};
        std::fprintf(stderr, "chad host profile, us per insert (%llu inserts):", (unsigned long long)calls);
        for (int k = 0; k < 9; k++) std::fprintf(stderr, " %s %.1f", names[k], us[k] / double(calls));
        std::fprintf(stderr, "\n");
    }
};
static HostProfile g_host_profile;
struct HostProfileScope {   // a whole call's time into one section
    int section; std::chrono::steady_clock::time_point t0;
    explicit HostProfileScope(int s) : section(s), t0(std::chrono::steady_clock::now()) {}
    ~HostProfileScope() { if (g_host_profile.on) g_host_profile.us[section] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
};

#define LAUNCH_TIMED(m_, id_, kernel_, grid_, block_, lds_, stream_, ...)                                                   \
    do {                                                                                                                    \
        if (hipEvent_t* ke_ = kernel_events(m_, id_)) hipExtLaunchKernelGGL(kernel_, grid_, block_, lds_, stream_, ke_[0], ke_[1], 0u, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel_, grid_, block_, lds_, stream_, __VA_ARGS__);                                        \
    } while (0)
inline bool times_anything(const chad_map* m) { return (m->flags & (CHAD_FLAG_TIMING | CHAD_FLAG_VERBOSE | CHAD_FLAG_TIME_UPDATE)) != 0; }

// Every hipEventRecord costs ~6 us of stream idle time, so CHAD_FLAG_TIME_UPDATE brackets only the update kernel.
// The reference's storage is effectively unbounded (VirtualArray reserves 4 GiB of address space per octree,
// virtual_array.hpp:16-19).  Here the pools double when the live block count -- mirrored by the device into host-mapped
// memory once per frame -- passes half the capacity: a new table twice the size, every brick re-inserted and moved.  Happens between frames; a single frame that outruns the remaining half still reports
// CHAD_ERR_CAPACITY.
chad_status grow_if_needed(chad_map* m) {
    if (m->fixed_capacity || *m->h_live_blocks < m->brick_capacity / 2 || m->brick_capacity >= (1u << 30)) return CHAD_OK;
    uint32_t live = 0;
    TRY(read_block_count(m, &live));
    if (live < m->brick_capacity / 2) return CHAD_OK;
    TRY(finish_closing(m, true));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->has_spare) {   // the next roll-over allocates one of the new size
        free_map_storage(m, m->spare); m->has_spare = false;
    }
    MapView bigger;
    const uint64_t blocks = std::min<uint64_t>(uint64_t(m->brick_capacity) * 2, 1ull << 30);
    const chad_status s = alloc_map_storage(m, bigger, blocks, m->stream);
    if (s != CHAD_OK) { free_map_storage(m, bigger); m->fixed_capacity = true; g_error.clear(); return CHAD_OK; }  // keep going at the old size
    if (live) hipLaunchKernelGGL(k_rehash_bricks, dim3(walk_grid(m->view.hash_mask)), dim3(256), 0, m->stream, m->view, bigger);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(bigger.overflow, m->view.overflow, 4, hipMemcpyDeviceToDevice, m->stream));   // (what the old table has reported stays reported)
    HIP_TRY(hipStreamSynchronize(m->stream));
    free_map_storage(m, m->view);
    m->view = bigger;
    m->brick_capacity = bigger.brick_capacity; m->hash_capacity = bigger.hash_mask + 1;
    *m->h_live_blocks = live;
    return CHAD_OK;
}

void record(chad_map* m, Stage s, hipStream_t stream) {
    if (times_stages(m) || ((m->flags & CHAD_FLAG_TIME_UPDATE) && (s == T_NORM || s == T_UPD)))
        (void)hipEventRecord(m->ev[m->timed_end % chad_map::kTimingRing][s], stream);
}

// mean stage times over the inserts whose event sets have not been read yet (the stream must be idle)
void collect_timing(chad_map* m) {
    if (!times_anything(m) || m->timed_end == m->timed_begin) return;
    double sum[T_COUNT] = {};
    int count = 0;
    for (uint64_t f = m->timed_begin; f < m->timed_end; f++) {
        hipEvent_t* e = m->ev[f % chad_map::kTimingRing];
        float t = 0;
        bool ok = true;
        if (!times_stages(m)) {  // update kernel only
            if (hipEventElapsedTime(&t, e[T_NORM], e[T_UPD]) == hipSuccess) { sum[T_UPD] += t; count++; } else (void)hipGetLastError();
            continue;
        }
        for (int s = T_CALC; s < T_COUNT; s++) { if (hipEventElapsedTime(&t, e[s - 1], e[s]) != hipSuccess) { ok = false; break; } sum[s] += t; }
        if (!ok || hipEventElapsedTime(&t, e[T_BEGIN], e[T_UPD]) != hipSuccess) { (void)hipGetLastError(); continue; }
        sum[T_BEGIN] += t;
        count++;
    }
    if (count) for (int s = 0; s < T_COUNT; s++) m->ms[s] = float(sum[s] / count);
    m->timed_begin = m->timed_end;
}

// The slot of the next insert, with room for n points.  Its buffers are free when its stream gets there: the frame that
// used the slot before (n_slots inserts ago) ran on the same stream, update included.
chad_status prepare_slot(chad_map* m, size_t n, FrameSlot** out) {
    const uint32_t turn = uint32_t(m->frames % m->n_slots);
    FrameSlot& f = m->slot[turn];
    // The lifetime contract of chad_map_insert_device: the frame that used this slot before (n_slots inserts ago) has read its
    // input for the last time before this insert returns.  The device says so itself (the launch behind the first sort pass
    // stores the frame's number into host-mapped memory: no event, nothing added to the stream); this also bounds how far the
    // host runs ahead of the device -- n_slots frames.  Normally the number is already there; otherwise poll briefly, then wait
    // for the slot's stream (always sufficient: everything of that frame ran on it).
    static const bool no_throttle = std::getenv("CHAD_DEBUG_NO_THROTTLE") != nullptr;   // measurement of the host's own cost per insert only: voids the lifetime contract
    if (f.input_seq && m->h_consumed[turn * 16] != f.input_seq && !no_throttle) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t polls = 0; m->h_consumed[turn * 16] != f.input_seq; polls++) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) { HIP_TRY(hipStreamSynchronize(f.prep)); break; }
            if (fold_busy(m) && (polls & 63u) == 63u) TRY(fold_advance(m, false));   // a roll-over's fold waits for the host once on its way: not for as long as the host waits here
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        m->host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    TRY(ensure_frame_capacity(m, f, n));
    if (m->caller_ready) {  // caller-supplied stream: the caller's work enqueued so far (it may produce the points) comes first
        HIP_TRY(hipEventRecord(m->caller_ready, m->stream));
        HIP_TRY(hipStreamWaitEvent(f.prep, m->caller_ready, 0));
    }
    m->last_slot = turn;
    *out = &f;
    return CHAD_OK;
}

// Submap bookkeeping of an insert, tsdf.cpp:46-61 (f32 distance in the reference's operation order; this file is built
// with -ffp-contract=off): the first pose of a submap is remembered, and a frame taken more than 5 m from it closes the
// submap and opens the next one.
// tsdf.cpp:50-52: would a frame taken at `position` finalize the active submap?
inline bool rolls_over(const chad_map* m, const float position[3]) {
    if (m->positions.empty()) return false;
    const std::array<float, 3>& first = m->positions.front();
    const float dx = position[0] - first[0], dy = position[1] - first[1], dz = position[2] - first[2];
    return std::sqrt(dx * dx + dy * dy + dz * dz) > 5.0f;
}

chad_status merge_and_close(chad_map* m, chad_comm* c);   // (further down)
// the active table has just been swapped for an empty one: nothing is staged in it yet (fold_if_due)
inline void staging_restart(chad_map* m) { m->pack_points = 0; for (auto& p : m->recent_points) p = 0; }

// the active submap has just been closed: its poses go with it
inline void retire_positions(chad_map* m) { m->closed_positions.push_back(std::move(m->positions)); m->positions.clear(); }
inline bool is_sharded(const chad_map* m) { return m->comm && m->comm->exchanges(); }

chad_status note_pose(chad_map* m, const float position[3]) {
    const std::array<float, 3> pose{position[0], position[1], position[2]};
    if (m->positions.empty()) m->positions.push_back(pose);
    else {
        if (rolls_over(m, position)) {
            HostSpan timed(m->pt.rollover_host_s);
            m->in_rollover_span = true;
            // every rank of a sharded map sees every pose, so all of them get here for the same frame: the merge is collective
            const chad_status closed = is_sharded(m) ? merge_and_close(m, m->comm) : close_active(m);
            m->in_rollover_span = false;
            TRY(closed);
            retire_positions(m);
            m->positions.push_back(pose);
        } else m->positions.push_back(pose);
    }
    return CHAD_OK;
}

// An insert without points is bookkeeping only (the frame-sharded multi-GPU map sends one to every rank that does not own
// the frame): nothing is enqueued.
chad_status insert_nothing(chad_map* m, const float position[3]) {
    TRY(note_pose(m, position));
    if (m->closing_comm) TRY(finish_closing(m, false));   // a rank that owns none of the coming frames still moves the exchange along
    m->frames_empty_tail++;
    m->last_points_in = 0;
    return CHAD_OK;
}

// The scatter form of the update stages its observations in one packed word per voxel whose count field holds 2^23 of them
// (map_kernels.hpp, kPackCountShift); k_fold_packed moves the staged part into the exact sums.  It is due when pack_fold_points(slots) points
// have been inserted since the last one -- a voxel receives at most one observation per point -- and only if some launch has used
// the staging words since (the device says so through a host-mapped word: a map fed dense frames never folds).  The frames that
// may not have run yet when the fold does are carried over: up to two per frame slot (a slot is reused once its previous frame's
// first sort pass has run, which says nothing about that frame's update).  The kernel runs on the inserting frame's stream, in
// front of its update, beside whatever the other streams are doing.
void fold_if_due(chad_map* m, hipStream_t stream, size_t n) {
    m->recent_points[m->recent_at++ % chad_map::kRecentFrames] = n;
    m->pack_points += n;
    if (m->pack_points < m->fold_due) return;
    const uint32_t seen = m->h_live_blocks[3];
    if (seen != m->pack_seen_folded) {
        hipLaunchKernelGGL(k_fold_packed, dim3(walk_grid(m->view.hash_mask)), dim3(256), 0, stream, m->view);
        m->pack_seen_folded = seen;
        m->folds++;
    }
    m->pack_points = 0;
    for (uint32_t k = 0; k < std::min<uint32_t>(2u * m->n_slots, chad_map::kRecentFrames); k++) m->pack_points += m->recent_points[(m->recent_at - 1u - k) % chad_map::kRecentFrames];
}

// tsdf.cpp:39-75 with the points already on the device (n > 0)
chad_status insert_device(chad_map* m, FrameSlot& f, const float* d_xyz, size_t n, const float position[3]) {
    if (n >= (1ull << 30)) return fail(CHAD_ERR_INVALID_ARG, "more than 2^30 points in one frame");
    g_host_profile.mark(8);
    TRY(note_pose(m, position));
    if (m->closing) {
        HostSpan timed(m->pt.rollover_host_s);
        m->in_rollover_span = true;
        const chad_status advanced = finish_closing(m, false);
        m->in_rollover_span = false;
        TRY(advanced);
    }   // a roll-over's archive, once its size is known (never waits)
    if (fold_busy(m)) TRY(fold_advance(m, false));   // ... and its fold into the DAG, a stage at a time (CHAD_FLAG_DAG_ONLY)
    m->frames_empty_tail = 0;

    TRY(grow_if_needed(m));
    const F3 pos{position[0], position[1], position[2]};
    const uint32_t count = uint32_t(n);
    // preparation on the slot's own stream (it has already waited for the slot's previous update: prepare_slot) ...
    hipStream_t prep = f.prep;
    record(m, T_BEGIN, prep);
    g_host_profile.mark(0);
    const uint32_t list_stride = uint32_t(f.cap_points / kListStripes + kGatherChunk);
    uint4* item_start = reinterpret_cast<uint4*>(f.d_seg + 2 * size_t(list_stride) * kListStripes);
    if (!count) hipLaunchKernelGGL(k_frame_init, dim3(1), dim3(256), 0, prep, f.d_hdr);
    if (count) {
        if (times_kernels(m) && m->ktimed_end - m->ktimed_begin >= uint64_t(chad_map::kKernelRing)) { TRY(join_streams(m)); collect_kernel_times(m); }
        // digit width of this frame's sort, from the key width the last finished frame reported (host-mapped): 9-bit digits where they
        // save a pass (17-18 key bits: two passes instead of three); the keys kernel derives the pass count from the same number
        static const uint32_t forced_bits = std::getenv("CHAD_DEBUG_DIGIT_BITS") ? uint32_t(std::atoi(std::getenv("CHAD_DEBUG_DIGIT_BITS"))) : 0u;
        const uint32_t key_bits_hint = uint32_t(m->h_live_blocks[2]);
        const uint32_t digit_bits = forced_bits == 8u || forced_bits == 9u ? forced_bits : (key_bits_hint ? radix_digit_bits(key_bits_hint) : uint32_t(kRadixBits));
        // as many single-pass kernels as recent frames needed (hint mirrored by the device into host-mapped memory), the
        // last of them covering whatever else the frame turns out to need (radix_sort.hpp, k_radix_onesweep to_limit)
        // launches: what a key of the reported width needs at this frame's digit width (all eight until a frame has reported)
        const uint32_t hint = m->sort_passes_forced ? m->sort_passes_forced
                              : (key_bits_hint ? std::min<uint32_t>(uint32_t(kMaxPasses), std::max<uint32_t>(1u, (key_bits_hint + digit_bits - 1u) / digit_bits))
                                               : std::min<uint32_t>(uint32_t(kMaxPasses), std::max<uint32_t>(1u, uint32_t(m->h_live_blocks[1]))));
        RadixSortSignals sig;
        sig.fault = m->d_overflow; sig.consumed = m->d_consumed + m->last_slot * 16u; sig.seq = ++f.input_seq;
        static const uint32_t big_block = std::getenv("CHAD_DEBUG_BIG_BLOCK") ? uint32_t(std::atoi(std::getenv("CHAD_DEBUG_BIG_BLOCK"))) : kBigBlock;
        static const uint32_t big_groups = std::getenv("CHAD_DEBUG_BIG_GROUPS") ? uint32_t(std::atoi(std::getenv("CHAD_DEBUG_BIG_GROUPS"))) : kNormalBigGroups;
        static const int normal_waves_per_cu = std::getenv("CHAD_DEBUG_NORMAL_WAVES") ? std::atoi(std::getenv("CHAD_DEBUG_NORMAL_WAVES")) : 20;
        const uint32_t normal_groups = std::max<uint32_t>(2u, m->cu_count * uint32_t(normal_waves_per_cu) / kNormalWaves);
        const float4* points0 = reinterpret_cast<const float4*>(f.sort.points[0]);
        const float4* points1 = reinterpret_cast<const float4*>(f.sort.points[1]);
        const dim3 gather_grid(std::min<uint32_t>(2048u, (count + kGatherChunk - 1) / kGatherChunk));
        const auto gather_kernel = digit_bits == uint32_t(kWideBits) ? k_frame_gather<kWideBits> : k_frame_gather<kRadixBits>;
        if (times_kernels(m) || times_stages(m)) {   // timing modes: every kernel (or stage) bracketed, one launch at a time
            LAUNCH_TIMED(m, CHAD_KERNEL_BOUNDS, k_frame_bounds, dim3(kFrameBlocks), dim3(256), 0, prep, d_xyz, count, pos, m->recip, f.d_part, f.sort.digit_base);
            LAUNCH_TIMED(m, CHAD_KERNEL_KEYS, k_frame_keys, dim3(kFrameBlocks), dim3(256), 0, prep, d_xyz, count, pos, m->recip, f.d_hdr, f.d_part, f.sort.keys[0],
                         f.sort.digit_base, f.sort.tile_status, uint32_t(radix_status_words(radix_tiles(count))), m->view.brick_count, m->d_live_blocks, digit_bits);
            record(m, T_CALC, prep);
            uint32_t pass_launches = 0;
            if (times_kernels(m)) { sig.pass_events = &m->kev[m->ktimed_end % chad_map::kKernelRing][2 * CHAD_KERNEL_SORT_PASS0]; sig.launches_out = &pass_launches; }
            RadixRest rest;
            radix_sort_points_enqueue(f.sort, d_xyz, count, kMaxPasses, &f.d_hdr->sort_passes, prep, kFrameBlocks, hint, sig, digit_bits, &rest);
            if (times_kernels(m)) m->kev_launched[m->ktimed_end % chad_map::kKernelRing] |= ((1u << pass_launches) - 1u) << CHAD_KERNEL_SORT_PASS0;
            LAUNCH_TIMED(m, CHAD_KERNEL_GATHER, gather_kernel, gather_grid, dim3(256), 0, prep, m->recip, f.d_hdr, f.d_part, (const uint64_t*)f.sort.keys[0], (const uint64_t*)f.sort.keys[1],
                         points0, points1, reinterpret_cast<uint2*>(f.d_seg), item_start, list_stride, sig.consumed, sig.seq, big_block, rest);
            record(m, T_SORT, prep);
            LAUNCH_TIMED(m, CHAD_KERNEL_NORMALS, k_frame_normals, dim3(normal_groups), dim3(64 * kNormalWaves), 0, prep, reinterpret_cast<float4*>(f.sort.points[0]), reinterpret_cast<float4*>(f.sort.points[1]), m->recip,
                         reinterpret_cast<const uint2*>(f.d_seg), list_stride, f.d_hdr, pos, f.d_normals, m->ablate, big_block, std::min(big_groups, normal_groups / 2u), item_start);
        } else {
            // the chain of preparation kernels, on the slot's stream
            hipLaunchKernelGGL(k_frame_bounds, dim3(kFrameBlocks), dim3(256), 0, prep, d_xyz, count, pos, m->recip, f.d_part, f.sort.digit_base);
            hipLaunchKernelGGL(k_frame_keys, dim3(kFrameBlocks), dim3(256), 0, prep, d_xyz, count, pos, m->recip, f.d_hdr, f.d_part, f.sort.keys[0], f.sort.digit_base, f.sort.tile_status,
                               uint32_t(radix_status_words(radix_tiles(count))), (const uint32_t*)m->view.brick_count, m->d_live_blocks, digit_bits);
            RadixRest rest;   // (the sort's catch-all, if its grid calls for one, is the gather kernel's prologue)
            radix_sort_points_enqueue(f.sort, d_xyz, count, kMaxPasses, &f.d_hdr->sort_passes, prep, kFrameBlocks, hint, sig, digit_bits, &rest);
            hipLaunchKernelGGL(gather_kernel, gather_grid, dim3(256), 0, prep, m->recip, f.d_hdr, (const FramePartial*)f.d_part, (const uint64_t*)f.sort.keys[0], (const uint64_t*)f.sort.keys[1], points0, points1,
                               reinterpret_cast<uint2*>(f.d_seg), item_start, list_stride, sig.consumed, sig.seq, big_block, rest);
            hipLaunchKernelGGL(k_frame_normals, dim3(normal_groups), dim3(64 * kNormalWaves), 0, prep, reinterpret_cast<float4*>(f.sort.points[0]), reinterpret_cast<float4*>(f.sort.points[1]), m->recip,
                               reinterpret_cast<const uint2*>(f.d_seg), list_stride, f.d_hdr, pos, f.d_normals, m->ablate, big_block, std::min(big_groups, normal_groups / 2u), (const uint4*)item_start);
        }
        if (times_stages(m)) record(m, T_NORM, prep);
    } else {
        record(m, T_CALC, prep); record(m, T_SORT, prep); if (times_stages(m)) record(m, T_NORM, prep);
    }
    g_host_profile.mark(3);
    // ... then the update, on the same stream.  It has to wait for whatever has needed the map to itself since this
    // stream last looked (a roll-over, growth ...), and a deferred reset is carried out here: behind the updates of the
    // other streams enqueued so far, in front of this one and -- through the epoch -- of every later one.
    if (f.seen_epoch != m->excl_epoch) { HIP_TRY(hipStreamWaitEvent(prep, m->excl_done, 0)); f.seen_epoch = m->excl_epoch; }
    if (m->pending_clear) {
        for (uint32_t k = 0; k < m->n_slots; k++) {
            FrameSlot& o = m->slot[k];
            if (&o == &f || !o.used || o.prep == prep) continue;
            hipEvent_t done;
            TRY(updates_done(o, &done));
            HIP_TRY(hipStreamWaitEvent(prep, done, 0));
        }
        enqueue_clear(m->view, prep);
        enqueue_flag_reset(m->view, prep, m->pending_clear_overflow);
        m->pending_clear = false; m->pending_clear_overflow = false;
        TRY(exclusive_end(m, prep));
        f.seen_epoch = m->excl_epoch;
    }
    g_host_profile.mark(4);
    if (count) {
        const float4* points0 = reinterpret_cast<const float4*>(f.sort.points[0]);
        const float4* points1 = reinterpret_cast<const float4*>(f.sort.points[1]);
        const bool adaptive = m->update_mode == CHAD_UPDATE_AUTO;
        uint32_t* pack_counter = m->d_overflow + 8, *pack_seen = m->d_live_blocks + 3;   // (the scatter form's "staging words in use" signal)
        if (m->update_mode != CHAD_UPDATE_TILED) fold_if_due(m, prep, n);
        // CHAD_FLAG_TIME_UPDATE times the update kernel alone, with the dispatch's own begin / end time stamps (hipExtLaunchKernel)
        const bool time_kernel = (!times_stages(m) && (m->flags & CHAD_FLAG_TIME_UPDATE)) || times_kernels(m);
        hipEvent_t* ev = m->ev[m->timed_end % chad_map::kTimingRing];
        hipEvent_t upd_pair[2] = {ev[T_NORM], ev[T_UPD]};
        if (hipEvent_t* ke = kernel_events(m, CHAD_KERNEL_UPDATE)) { upd_pair[0] = ke[0]; upd_pair[1] = ke[1]; }
        if (m->update_mode == CHAD_UPDATE_SCATTER) {
            const dim3 grid = grid_for(count);
            if (time_kernel)
                hipExtLaunchKernelGGL(k_update_scatter, grid, dim3(256), 0u, prep, upd_pair[0], upd_pair[1], 0u, points0, points1, f.d_normals, f.d_hdr, pos, m->res, m->recip, m->trunc, m->view, pack_counter, pack_seen);
            else
                hipLaunchKernelGGL(k_update_scatter, grid, dim3(256), 0, prep, points0, points1, f.d_normals, f.d_hdr, pos, m->res, m->recip, m->trunc, m->view, pack_counter, pack_seen);
        } else  // the tiled kernel; in automatic mode it takes the scatter form itself when the frame is sparse (decided on the device)
            tile_update_enqueue(m->tile, points0, points1, f.d_normals, item_start, f.d_hdr, pos, m->res, m->recip, m->trunc, m->view, prep, adaptive, list_stride, pack_counter, pack_seen, m->d_stamps,
                                time_kernel ? upd_pair[0] : nullptr, time_kernel ? upd_pair[1] : nullptr);
        if (times_stages(m)) record(m, T_UPD, prep);
    } else {
        if (!times_stages(m)) record(m, T_NORM, prep);
        record(m, T_UPD, prep);
    }
    g_host_profile.mark(5);
    if (count && (m->flags & CHAD_FLAG_SPACE_CARVING))   // the extension of SURVEY.md section 8 f4: free space along every ray, behind the update (frames run one at a time)
        hipLaunchKernelGGL(k_carve, grid_for(count, 256, 4096), dim3(256), 0, prep, reinterpret_cast<const float4*>(f.sort.points[0]), reinterpret_cast<const float4*>(f.sort.points[1]),
                           f.d_hdr, pos, m->res, m->recip, m->trunc, m->view);
    f.tail_unrecorded = true;   // (update_done is recorded when somebody waits for it: updates_done)
    f.used = true;
    HIP_TRY(hipGetLastError());
    m->frames++;
    m->last_points_in = n;
    if (times_kernels(m)) m->ktimed_end++;
    if (times_anything(m)) {
        m->timed_end++;
        if (m->timed_end - m->timed_begin > chad_map::kTimingRing) m->timed_begin = m->timed_end - chad_map::kTimingRing;
        if (m->flags & CHAD_FLAG_VERBOSE) {  // the reference prints its five stage lines on every insert
            TRY(join_streams(m));
            collect_timing(m);
            std::printf("mc  calc %.2f\nmc  sort %.2f\nnorm est %.2f\noct  upd %.2f\ntotal    %.2f\n", m->ms[T_CALC], m->ms[T_SORT], m->ms[T_NORM], m->ms[T_UPD], m->ms[T_BEGIN]);
        }
    }
    g_host_profile.mark(6);
    return CHAD_OK;
}

}  // namespace

extern "C" {

const char* chad_last_error(void) { return g_error.c_str(); }
const char* chad_version(void) { return "chad_hip 0.1 gfx950"; }

chad_status chad_map_create(float sdf_res, float sdf_trunc, const chad_options* opt, chad_map** out) {
    if (!out) return fail(CHAD_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!(sdf_res > 0.0f) || !(sdf_trunc > 0.0f) || !std::isfinite(sdf_res) || !std::isfinite(sdf_trunc))
        return fail(CHAD_ERR_INVALID_ARG, "sdf_res and sdf_trunc must be positive and finite");
    chad_options o{};
    o.device = -1;
    if (opt) {
        if (opt->struct_size < sizeof(uint32_t) * 2) return fail(CHAD_ERR_INVALID_ARG, "chad_options.struct_size not set");
        std::memcpy(&o, opt, std::min<size_t>(opt->struct_size, sizeof(o)));
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return fail(CHAD_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    }
    int dev = o.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return fail(CHAD_ERR_NO_DEVICE, "device ordinal out of range");
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(CHAD_ERR_NO_DEVICE, "cannot select the requested device");

    chad_map* m = new chad_map();
    m->res = sdf_res; m->trunc = sdf_trunc;
    m->recip = float(1.0 / double(sdf_res));  // morton.hpp:63, octree.hpp:82
    m->device = dev; m->flags = o.flags;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) m->cu_count = uint32_t(cus);
        else (void)hipGetLastError();
        m->tile = tile_launch_shape(sdf_res, sdf_trunc, m->cu_count);
    }
    m->update_mode = o.update_mode <= CHAD_UPDATE_TILED ? o.update_mode : CHAD_UPDATE_AUTO;
    m->fixed_capacity = (o.flags & CHAD_FLAG_FIXED_CAPACITY) != 0;
    m->dag_on_host = std::getenv("CHAD_DEBUG_DAG_HOST") != nullptr;
    if (const char* e = std::getenv("CHAD_STAGE_MODE")) m->stage_mode = !std::strcmp(e, "staged") ? 1 : !std::strcmp(e, "direct") ? 2 : 0;
    if (const char* e = std::getenv("CHAD_STAGE_DIRECT_KB")) m->stage_direct_from = size_t(std::max(0, std::atoi(e))) << 10;
    if (const char* e = std::getenv("CHAD_DEBUG_FAIL_FOLD")) m->debug_fail_fold = std::atoi(e);
    if (const char* e = std::getenv("CHAD_DEBUG_FAIL_CLOSE")) m->debug_fail_close = std::atoi(e);
    if (const char* a = std::getenv("CHAD_DEBUG_ABLATE")) m->ablate = uint32_t(std::strtoul(a, nullptr, 0));
    if (const char* a = std::getenv("CHAD_DEBUG_FOLD_POINTS")) { m->fold_due = uint64_t(std::atoll(a)); m->report_folds = true; }
    if (std::getenv("CHAD_DEBUG_STAMPS")) { (void)chad_malloc(&m->d_stamps, 8192 * 16 * 8); (void)hipMemset(m->d_stamps, 0, 8192 * 16 * 8); }
    auto bail = [&](chad_status s) { chad_map_destroy(m); return s; };
#define CREATE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); \
        return bail(fail(e_ == hipErrorOutOfMemory ? CHAD_ERR_OUT_OF_MEMORY : CHAD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_))); } } while (0)
    if (o.stream) m->stream = hipStream_t(o.stream);
    else {
        // The map's own stream carries what everything else waits for -- the clear behind a reset, a roll-over's table swap, the
        // archive and the exchange -- and gets the highest priority: at the default priority those queue behind the frames' kernels
        // (0.132 instead of 0.094 ms per 1 M-point frame, DESIGN.md dead ends).  The frames' own streams (preparation AND update) stay
        // at the default: priorities between streams that depend on each other invert (a collapse at four slots).
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (std::getenv("CHAD_DEBUG_NO_PRIORITY")) CREATE_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
        else CREATE_TRY(hipStreamCreateWithPriority(&m->stream, hipStreamNonBlocking, greatest));
        m->own_stream = true;
    }
    if (const char* e = std::getenv("CHAD_SLOTS")) m->n_slots = std::min<uint32_t>(chad_map::kMaxSlots, std::max(1, std::atoi(e)));
    if (!m->report_folds) m->fold_due = pack_fold_points(m->n_slots);   // (unless CHAD_DEBUG_FOLD_POINTS has set it)
    m->serial = (m->flags & (CHAD_FLAG_TIMING | CHAD_FLAG_VERBOSE | CHAD_FLAG_SPACE_CARVING)) != 0 || std::getenv("CHAD_DEBUG_SERIAL") != nullptr ||
                ((m->flags & CHAD_FLAG_TIME_KERNELS) && !std::getenv("CHAD_DEBUG_TIME_PIPELINED"));   // (experiment: per-kernel time stamps with three frames in flight)

    uint64_t blocks = o.max_blocks ? o.max_blocks : (1ull << 21);
    if (blocks > (1ull << 30)) return bail(fail(CHAD_ERR_INVALID_ARG, "max_blocks above 2^30"));
    {   // the DAG's memory, reserved up front (alloc.hpp): levels then grow without a hipMalloc; CHAD_DAG_POOL_MB overrides
        uint64_t pool = o.dag_pool_bytes;
        if (const char* e = std::getenv("CHAD_DAG_POOL_MB")) pool = uint64_t(std::max(0ll, std::atoll(e))) << 20;
        if (pool) CREATE_TRY(m->ddag.pool.reserve(pool));
    }
    CREATE_TRY(chad_malloc(&m->d_overflow, 64));   // 16 words: [0] sort faults, [1..7] the tables' flags, [8] staging-word launches, [9..15] the tables' clear counters
    CREATE_TRY(hipMemsetAsync(m->d_overflow, 0, 64, m->stream));
    if (alloc_map_storage(m, m->view, blocks, m->stream) != CHAD_OK) return bail(CHAD_ERR_OUT_OF_MEMORY);
    m->brick_capacity = m->view.brick_capacity; m->hash_capacity = m->view.hash_mask + 1;
    for (uint32_t k = 0; k < m->n_slots; k++) {
        FrameSlot& f = m->slot[k];
        if (m->serial) f.prep = m->stream;
        else CREATE_TRY(hipStreamCreateWithFlags(&f.prep, hipStreamNonBlocking));
        CREATE_TRY(hipEventCreateWithFlags(&f.update_done, hipEventDisableTiming));
        CREATE_TRY(hipEventCreateWithFlags(&f.stage_done, hipEventDisableTiming));
        CREATE_TRY(chad_malloc(&f.d_hdr, sizeof(FrameHeader)));
        CREATE_TRY(hipMemsetAsync(f.d_hdr, 0, sizeof(FrameHeader), m->stream));
        CREATE_TRY(chad_malloc(&f.d_part, sizeof(FramePartial) * kFrameBlocks));
    }
    CREATE_TRY(hipEventCreateWithFlags(&m->excl_done, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&m->closing_counted, hipEventDisableTiming));
    CREATE_TRY(hipHostMalloc(&m->closing_count, kCountStripes * 64 + 64));   // (the counter stripes, and behind them the closing table's flag word)
    m->closing_count[kCountStripes * 16] = 0u;
    if (!m->own_stream) CREATE_TRY(hipEventCreateWithFlags(&m->caller_ready, hipEventDisableTiming));
    CREATE_TRY(hipHostMalloc(&m->h_hdr, sizeof(FrameHeader)));
    CREATE_TRY(hipHostMalloc(&m->h_live_blocks, 64, hipHostMallocMapped));
    m->h_live_blocks[0] = 0;
    m->h_live_blocks[1] = kMaxPasses;  // sort pass hint: all passes as single launches until the first frame has reported
    m->h_live_blocks[2] = 0;           // key width hint: none yet (8-bit digits)
    m->h_live_blocks[3] = 0;           // launches that have used the scatter form's staging words (fold_if_due)
    if (const char* e = std::getenv("CHAD_DEBUG_SORT_PASSES")) m->sort_passes_forced = uint32_t(std::atoi(e));
    CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&m->d_live_blocks), const_cast<uint32_t*>(m->h_live_blocks), 0));
    CREATE_TRY(hipHostMalloc(&m->h_consumed, chad_map::kMaxSlots * 64, hipHostMallocMapped));
    std::memset(const_cast<uint32_t*>(m->h_consumed), 0, chad_map::kMaxSlots * 64);
    CREATE_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&m->d_consumed), const_cast<uint32_t*>(m->h_consumed), 0));
    CREATE_TRY(hipEventCreateWithFlags(&m->producer_ready, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&m->merge_event, hipEventDisableTiming));
    CREATE_TRY(hipEventCreate(&m->fold_t0)); CREATE_TRY(hipEventCreate(&m->fold_t1));
    if (m->flags & CHAD_FLAG_DAG_ONLY) for (int k = 0; k < 2; k++) { CREATE_TRY(hipEventCreate(&m->close_t0[k])); CREATE_TRY(hipEventCreate(&m->close_t1[k])); }
    if (m->serial || !(m->flags & CHAD_FLAG_DAG_ONLY)) m->fold_stream = nullptr;   // (only CHAD_FLAG_DAG_ONLY roll-overs use the two streams: a map without the flag does not hold hardware queues for them)
    else if (std::getenv("CHAD_DEBUG_FOLD_NO_PRIORITY")) CREATE_TRY(hipStreamCreateWithFlags(&m->fold_stream, hipStreamNonBlocking));
    else {
        // Streams of one priority share a few hardware queues: at the default priority the fold's chain of fifty launches sat in the
        // queue of one of the frame slots, behind every frame that slot took (two 0.3 ms chains inside one 3.5 ms fold, rocprofv3
        // --kernel-trace), and the next roll-over waits for the fold.  At the highest priority it shares with the map's stream only.
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        CREATE_TRY(hipStreamCreateWithPriority(&m->fold_stream, hipStreamNonBlocking, greatest));
    }
    if (m->fold_stream) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        CREATE_TRY(hipStreamCreateWithPriority(&m->close_stream, hipStreamNonBlocking, std::getenv("CHAD_DEBUG_FOLD_NO_PRIORITY") ? least : greatest));
    }
    CREATE_TRY(hipEventCreateWithFlags(&m->clusters_ready, hipEventDisableTiming));
    CREATE_TRY(hipHostMalloc(&m->merge_host, chad_map::kMergeHostWords * 8));
    // what the look-back sort may assume about residency: the pass kernel's real occupancy, shared among the frames in flight
    m->sort_resident = radix_resident_workgroups<PointPayload>();
    m->aux_resident = radix_resident_workgroups<IndexPayload>() / (m->serial ? 1u : m->n_slots + 1u);
    m->aux_sort.resident = m->aux_resident;
    if (std::getenv("CHAD_DEBUG_SORT_TICKETS")) m->sort_resident = 0u;   // experiments: every sort takes tickets
    for (uint32_t k = 0; k < m->n_slots; k++) m->slot[k].sort.resident = m->sort_resident / (m->serial ? 1u : m->n_slots);
    {   // the catch-all of a frame's sort runs inside k_frame_gather: its participants must be resident THERE (that kernel's occupancy, three
        // quarters of it, shared among the frames in flight; its grid is at most 2 048 workgroups) -- ADVICE r3
        int per_cu8 = 0, per_cu9 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu8, k_frame_gather<kRadixBits>, 256, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu9, k_frame_gather<kWideBits>, 256, 0) != hipSuccess) { (void)hipGetLastError(); per_cu8 = per_cu9 = 0; }
        const uint32_t hosts = std::min<uint32_t>(2048u, uint32_t(std::max(0, std::min(per_cu8, per_cu9))) * m->cu_count * 3u / 4u) / (m->serial ? 1u : m->n_slots);
        for (uint32_t k = 0; k < m->n_slots; k++) m->slot[k].sort.rest_limit = hosts;
    }
    CREATE_TRY(chad_malloc(&m->d_cursor, 8));   // ([1]: k_table_keys' cursor on close_stream, beside whatever the map's stream compacts)
    if (times_anything(m))
        for (auto& set : m->ev) for (auto& e : set) CREATE_TRY(hipEventCreate(&e));
    if (times_kernels(m))
        for (auto& set : m->kev) for (auto& e : set) CREATE_TRY(hipEventCreate(&e));
    for (uint32_t k = 0; k < m->n_slots; k++)
        if (ensure_frame_capacity(m, m->slot[k], o.max_points ? o.max_points : (1u << 20)) != CHAD_OK) return bail(CHAD_ERR_OUT_OF_MEMORY);
    CREATE_TRY(hipStreamSynchronize(m->stream));
#undef CREATE_TRY
    *out = m;
    return CHAD_OK;
}

static void orphan_iterators(chad_map* m);   // (chad_iter is defined with the read-back API below)

void chad_map_destroy(chad_map* m) {
    if (!m) return;
    orphan_iterators(m);   // iterators that outlive the map report invalid instead of dangling
    DeviceGuard guard(m->device);
    if (m->closing_comm) (void)finish_closing(m, true);   // an exchange in flight is collective: this rank's remaining stages are issued before the map goes
    if (m->close_stream) (void)hipStreamSynchronize(m->close_stream);
    if (m->fold_stream) (void)hipStreamSynchronize(m->fold_stream);
    for (auto& f : m->slot) if (f.prep && f.prep != m->stream) (void)hipStreamSynchronize(f.prep);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    g_host_profile.report();
    if (m->report_folds) std::fprintf(stderr, "chad: staging folds %llu\n", (unsigned long long)m->folds);
    if (m->d_stamps) {  // diagnostic: mean cycles per phase over the waves of the last update launch
        std::vector<unsigned long long> st(8192 * 16);
        (void)hipMemcpy(st.data(), m->d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> life, barriers, items, entry, loop, done, zeroed, lists;
        for (int w = 0; w < 8192; w++) {
            if (!st[w * 16 + 5]) continue;
            life.push_back(st[w * 16 + 5]); barriers.push_back(st[w * 16 + 1]); items.push_back(st[w * 16 + 2]);
            entry.push_back(st[w * 16 + 8]); loop.push_back(st[w * 16 + 9]); done.push_back(st[w * 16 + 10]); zeroed.push_back(st[w * 16 + 11]); lists.push_back(st[w * 16 + 12]);
        }
        if (!life.empty()) {
            auto pct = [](std::vector<unsigned long long> v, double p) { std::sort(v.begin(), v.end()); return v[size_t(p * (v.size() - 1))]; };
            std::fprintf(stderr, "stamps: waves %zu  life p10 %llu p50 %llu p90 %llu max %llu cycles | barriers p50 %llu p90 %llu max %llu | items p10 %llu p50 %llu p90 %llu max %llu\n", life.size(),
                         pct(life, 0.1), pct(life, 0.5), pct(life, 0.9), pct(life, 1.0), pct(barriers, 0.5), pct(barriers, 0.9), pct(barriers, 1.0),
                         pct(items, 0.1), pct(items, 0.5), pct(items, 0.9), pct(items, 1.0));
            // the constant clock (10 ns ticks), relative to the first wave's entry: when the waves were started, reached their first item and finished
            const unsigned long long e0 = pct(entry, 0.0);
            auto rel = [&](std::vector<unsigned long long> v, double p) { return 0.01 * double(pct(v, p) - e0); };
            std::fprintf(stderr, "stamps: us after the first wave's entry: entry p50 %.2f p90 %.2f max %.2f | first item p10 %.2f p50 %.2f p90 %.2f max %.2f | done p10 %.2f p50 %.2f p90 %.2f max %.2f\n",
                         rel(entry, 0.5), rel(entry, 0.9), rel(entry, 1.0), rel(loop, 0.1), rel(loop, 0.5), rel(loop, 0.9), rel(loop, 1.0), rel(done, 0.1), rel(done, 0.5), rel(done, 0.9), rel(done, 1.0));
            std::fprintf(stderr, "stamps: us after the first wave's entry: tile zeroed p50 %.2f p90 %.2f | lists in LDS p50 %.2f p90 %.2f max %.2f\n", rel(zeroed, 0.5), rel(zeroed, 0.9), rel(lists, 0.5), rel(lists, 0.9), rel(lists, 1.0));
            double sum[8] = {};
            for (int w = 0; w < 8192; w++) if (st[w * 16 + 5]) for (int k = 0; k < 8; k++) sum[k] += double(st[w * 16 + k]);
            std::fprintf(stderr, "stamps: mean cycles per wave: fetch %.0f walk %.0f readback %.0f hash %.0f flush %.0f life %.0f\n", sum[0] / life.size(), sum[3] / life.size(),
                         sum[6] / life.size(), sum[7] / life.size(), sum[4] / life.size(), sum[5] / life.size());
        }
        (void)chad_free(m->d_stamps);
    }
    for (auto& f : m->slot) {
        free_frame_buffers(f);
        (void)chad_free(f.d_hdr); (void)chad_free(f.d_part);
        if (f.update_done) (void)hipEventDestroy(f.update_done);
        if (f.stage_done) (void)hipEventDestroy(f.stage_done);
        if (f.prep && f.prep != m->stream) (void)hipStreamDestroy(f.prep);
    }
    if (m->caller_ready) (void)hipEventDestroy(m->caller_ready);
    if (m->producer_ready) (void)hipEventDestroy(m->producer_ready);
    if (m->merge_event) (void)hipEventDestroy(m->merge_event);
    if (m->merge_host) (void)hipHostFree(m->merge_host);
    m->merge_a.release(); m->merge_b.release(); m->merge_c.release(); m->archive_arena.release(); m->cluster_arena.release(); m->cluster_arenas[0].release(); m->cluster_arenas[1].release();
    if (m->h_consumed) (void)hipHostFree(const_cast<uint32_t*>(m->h_consumed));
    free_sort_buffers(m->aux_sort);
    m->ddag.release(true);
    if (m->fold.ready) (void)hipEventDestroy(m->fold.ready);
    if (m->fold_t0) (void)hipEventDestroy(m->fold_t0);
    if (m->fold_t1) (void)hipEventDestroy(m->fold_t1);
    for (int k = 0; k < 2; k++) { if (m->close_t0[k]) (void)hipEventDestroy(m->close_t0[k]); if (m->close_t1[k]) (void)hipEventDestroy(m->close_t1[k]); }
    if (m->fold_stream) (void)hipStreamDestroy(m->fold_stream);
    if (m->close_stream) (void)hipStreamDestroy(m->close_stream);
    if (m->clusters_ready) (void)hipEventDestroy(m->clusters_ready);
    if (m->excl_done) (void)hipEventDestroy(m->excl_done);
    if (m->closing_counted) (void)hipEventDestroy(m->closing_counted);
    if (m->closing_count) (void)hipHostFree(m->closing_count);
    if (m->closing) free_map_storage(m, m->closing_view);
    if (m->has_spare) free_map_storage(m, m->spare);
    for (auto& s : m->closed) free_submap(s);
    free_submap(m->active_copy);
    free_map_storage(m, m->view);
    (void)chad_free(m->d_overflow); (void)chad_free(m->d_cursor);
    if (m->h_live_blocks) (void)hipHostFree(const_cast<uint32_t*>(m->h_live_blocks));
    if (m->h_hdr) (void)hipHostFree(m->h_hdr);
    for (auto& set : m->ev) for (auto& e : set) if (e) (void)hipEventDestroy(e);
    for (auto& set : m->kev) for (auto& e : set) if (e) (void)hipEventDestroy(e);
    if (m->own_stream && m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

struct InsertTimer {   // host time of one insert call into chad_map::host_insert_s
    chad_map* m; std::chrono::steady_clock::time_point t0;
    explicit InsertTimer(chad_map* map) : m(map), t0(std::chrono::steady_clock::now()) {}
    ~InsertTimer() { if (m) { m->host_insert_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); m->host_inserts++; } }
};

// frame-sharded map: is insert number stream_frames (counted from the attach) this rank's?  Counts the insert.
static bool owns_next_frame(chad_map* m) { return int32_t(m->stream_frames++ % uint64_t(m->comm->world)) == m->comm->rank; }

chad_status chad_map_insert_device(chad_map* m, const float* xyz_device, size_t n, const float position[3]) {
    g_host_profile.begin();
    if (!m || !position || (n && !xyz_device)) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    InsertTimer timed(m);
    DeviceGuard guard(m->device);
    if (is_sharded(m) && !owns_next_frame(m)) n = 0;   // another rank integrates this frame; the pose is recorded here too
    if (!n) return insert_nothing(m, position);
    FrameSlot* f = nullptr;
    TRY(prepare_slot(m, n, &f));
    return insert_device(m, *f, xyz_device, n, position);
}

chad_status chad_map_insert_device_stream(chad_map* m, const float* xyz_device, size_t n, const float position[3], void* producer_stream) {
    g_host_profile.begin();
    if (!m || !position || (n && !xyz_device)) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    InsertTimer timed(m);
    DeviceGuard guard(m->device);
    if (is_sharded(m) && !owns_next_frame(m)) n = 0;   // another rank integrates this frame; the pose is recorded here too
    if (!n) return insert_nothing(m, position);
    FrameSlot* f = nullptr;
    TRY(prepare_slot(m, n, &f));
    // whatever the producer's stream has been given so far -- the kernels or copies that write the points -- comes first.  A
    // stream that is idle right now has nothing the frame could overtake (a query costs a microsecond; record + wait cost twelve)
    if (hipStreamQuery(hipStream_t(producer_stream)) != hipSuccess) {
        (void)hipGetLastError();   // hipErrorNotReady
        HIP_TRY(hipEventRecord(m->producer_ready, hipStream_t(producer_stream)));
        HIP_TRY(hipStreamWaitEvent(f->prep, m->producer_ready, 0));
    }
    return insert_device(m, *f, xyz_device, n, position);
}

chad_status chad_map_insert(chad_map* m, const float* xyz, size_t n, const float position[3]) {
    if (!m || !position || (n && !xyz)) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    InsertTimer timed(m);
    DeviceGuard guard(m->device);
    if (is_sharded(m) && !owns_next_frame(m)) n = 0;   // another rank integrates this frame; the pose is recorded here too
    if (!n) return insert_nothing(m, position);
    FrameSlot* fp = nullptr;
    TRY(prepare_slot(m, n, &fp));
    FrameSlot& f = *fp;
    // The caller's buffer is borrowed for the duration of the call only (tsdf.hpp:50-57 even copies it), and it is pageable memory.
    // Two ways into HBM (scripts/microbench_h2d.hip, 12 MB on an EPYC 9575F box: the pinned-memory ceiling is 0.232 ms):
    //   direct  hipMemcpy from the caller's pointer: the runtime page-locks the range in place and the DMA engine reads it there --
    //           0.231 ms, the ceiling itself, and no CPU copy whose speed depends on where the box's memory sits (the staged path
    //           needed 0.33-2.35 ms over the boxes seen).  The call returns when the bytes are in HBM: the buffer is the caller's
    //           again, and meanwhile the device works on the frames before this one (their kernels need less than the copy).
    //   staged  memcpy into the slot's pinned buffer in chunks, every chunk's DMA behind it (CHAD_STAGE_MODE=staged; and for small
    //           frames, where a memcpy of a few hundred KB beats the runtime's page-locking)
    const bool direct = m->stage_mode == 2 || (m->stage_mode == 0 && n * 12 >= m->stage_direct_from);
    if (direct) {
        // f.d_xyz is free: the frame that used this slot before has read its input for the last time (prepare_slot waited for the device's word)
        const auto t0 = std::chrono::steady_clock::now();
        HIP_TRY(hipMemcpy(f.d_xyz, xyz, n * 12, hipMemcpyHostToDevice));
        m->host_copy_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } else {
        if (!f.h_stage) HIP_TRY(hipHostMalloc(&f.h_stage, f.cap_points * 12, hipHostMallocDefault));
        else HIP_TRY(hipEventSynchronize(f.stage_done));  // its previous frame (n_slots inserts ago) has left the buffer
        static const size_t chunk = std::getenv("CHAD_STAGE_CHUNK_KB") ? size_t(std::atoi(std::getenv("CHAD_STAGE_CHUNK_KB"))) << 10 : size_t(2) << 20;  // bytes per chunk (scripts/host_insert_sweep.py)
        static const int helpers = std::getenv("CHAD_STAGE_THREADS") ? std::max(0, std::atoi(std::getenv("CHAD_STAGE_THREADS")) - 1) : 1;   // one helper beside the caller: 0.57 -> 0.40 ms per 12 MB frame; a third thread gains nothing
        const char* src = reinterpret_cast<const char*>(xyz);
        char* pinned = reinterpret_cast<char*>(f.h_stage);
        char* device = reinterpret_cast<char*>(f.d_xyz);
        hipError_t dma = hipSuccess;
        auto start_dma = [&](size_t at, size_t len) { if (dma == hipSuccess) dma = hipMemcpyAsync(device + at, pinned + at, len, hipMemcpyHostToDevice, f.prep); };
        if (helpers > 0 && n * 12 >= 3 * chunk) {   // a helper thread shares the copy
            if (!m->copy_pool && !m->copy_pool_failed) {
                try { m->copy_pool.reset(new CopyPool(helpers)); } catch (...) { m->copy_pool_failed = true; }   // (no thread to be had: the caller copies alone)
            }
        }
        if (m->copy_pool && n * 12 >= 3 * chunk) m->copy_pool->copy(src, pinned, n * 12, chunk, start_dma);
        else
            for (size_t at = 0; at < n * 12; at += chunk) {
                const size_t len = std::min(chunk, n * 12 - at);
                std::memcpy(pinned + at, src + at, len);
                start_dma(at, len);
            }
        HIP_TRY(dma);
        HIP_TRY(hipEventRecord(f.stage_done, f.prep));
    }
    return insert_device(m, f, f.d_xyz, n, position);
}

// the last insert's header (stage read-back, statistics) and the sticky overflow flag, after everything has finished
chad_status read_back_header(chad_map* m, uint32_t* overflow) {
    TRY(flush_pending_clear(m));
    TRY(join_streams(m));
    HIP_TRY(hipMemcpy(m->h_hdr, m->slot[m->last_slot].d_hdr, offsetof(FrameHeader, first_item), hipMemcpyDeviceToHost));   // (the scalars and counters; not the update kernel's item table)
    uint32_t words[8];
    HIP_TRY(hipMemcpy(words, m->d_overflow, sizeof(words), hipMemcpyDeviceToHost));
    // the sort's fault bits, and the overflow bit of the tables that hold data: the active one and one that is closing (a spare table's
    // word is stale until the table is swapped in: spare_ready)
    *overflow = words[0] | m->overflow_latched;
    if (m->view.overflow) *overflow |= words[m->view.overflow - m->d_overflow] & kTableOverflowed;
    if (m->closing && m->closing_view.overflow) *overflow |= words[m->closing_view.overflow - m->d_overflow] & kTableOverflowed;
    return CHAD_OK;
}

chad_status chad_map_sync(chad_map* m) {
    if (!m) return fail(CHAD_ERR_INVALID_ARG, "null map");
    DeviceGuard guard(m->device);
    uint32_t overflow = 0;
    TRY(finish_closing(m, true));   // a roll-over's archive (or exchange: every rank syncs at the same point) is part of "all enqueued work"
    while (fold_busy(m)) TRY(fold_advance(m, true));   // ... and so is its fold into the DAG (CHAD_FLAG_DAG_ONLY)
    TRY(read_back_header(m, &overflow));
    if (!m->ddag.outgrown.empty()) { TRY(join_streams(m)); m->ddag.free_outgrown(); }   // what the DAG's levels have grown out of: freed where a hipFree's device-wide wait costs nothing
    if (overflow & kSortFaultStalled) (void)hipMemset(m->d_overflow, 0, 4);   // reported once (every stream is idle here); capacity flags stay until a reset
    if (overflow & kSortFaultStalled) return fail(CHAD_ERR_HIP, "a radix sort tile gave up waiting for its predecessors (workgroups not dispatched in index order?): at least one frame was sorted incompletely");
    if (overflow) return fail(CHAD_ERR_CAPACITY, "block capacity of the active submap exceeded; raise chad_options.max_blocks");
    return CHAD_OK;
}

chad_status chad_map_reset(chad_map* m) {
    if (!m) return fail(CHAD_ERR_INVALID_ARG, "null map");
    HostProfileScope profiled(7);
    DeviceGuard guard(m->device);
    TRY(finish_closing(m, true));
    while (fold_busy(m)) TRY(fold_advance(m, true));
    if (!m->closed.empty()) {
        TRY(join_streams(m));
        for (auto& s : m->closed) free_submap(s);
        m->closed.clear();
    }
    m->overflow_latched = 0;
    m->dag = Dag();
    m->ddag.release();   // (the levels' stores go back to the pool; the reservation itself -- chad_options.dag_pool_bytes -- stays)
    m->dag_broken = false; m->dag_remote = false;
    m->dag_mirror_stale = false;
    m->dag_roots.clear();
    m->dag_generation++;
    m->mesh_submap = INT32_MIN;
    m->positions.clear(); m->closed_positions.clear(); m->closed_pose.clear();
    // Octree::clear() by rotating tables, as a roll-over does: the empty spare table becomes the active one at once, and the old
    // one is cleared on the map's stream once the updates enqueued so far have landed in it -- beside the next frames, which no
    // longer wait for them (a clear in front of the next update made every step's update wait for the two before it: 0.113 ->
    // 0.10x ms per step of the bench, which resets before every insert).  The first reset of a map pays for the spare table.
    static const bool no_rotation = std::getenv("CHAD_DEBUG_NO_RESET_ROTATION") != nullptr;
    if (!m->pending_clear && !no_rotation) {
        if (!m->has_spare && !m->spare_unaffordable) {
            if (alloc_map_storage(m, m->spare, m->brick_capacity, m->stream) == CHAD_OK) m->has_spare = true;
            else { free_map_storage(m, m->spare); m->spare_unaffordable = true; g_error.clear(); }
        }
        if (m->has_spare) {
            TRY(spare_ready(m, true));   // (incl. the spare table's flag word: no frame is using that table)
            TRY(exclusive_end(m, m->stream));   // later updates wait for the map's stream up to HERE: the spare's own clear lies before this point
            TRY(exclusive_begin(m));            // ... and the old table's clear waits for every update enqueued so far, nobody waits for it
            enqueue_clear(m->view, m->stream, *m->h_live_blocks);   // (its flag word is reset when the table is swapped in again: spare_ready / the memset above)
            HIP_TRY(hipGetLastError());
            std::swap(m->view, m->spare);
            m->spare_flags = 2;
            *m->h_live_blocks = 0;
            staging_restart(m);
            return CHAD_OK;
        }
    }
    // no room for a second table: the clear is deferred to the stream of the next insert (right in front of its update), so that
    // the next frame's preparation does not queue behind it; anything else that looks at the map first carries it out (flush_pending_clear)
    m->pending_clear = true;
    m->pending_clear_overflow = true;
    return CHAD_OK;
}

chad_status chad_map_would_roll_over(chad_map* m, const float position[3], int32_t* yes) {
    if (!m || !position || !yes) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    *yes = rolls_over(m, position) ? 1 : 0;
    return CHAD_OK;
}

chad_status chad_map_close_submap(chad_map* m) {
    if (!m) return fail(CHAD_ERR_INVALID_ARG, "null map");
    DeviceGuard guard(m->device);
    if (m->positions.empty()) return CHAD_OK;
    TRY(is_sharded(m) ? merge_and_close(m, m->comm) : close_active(m));
    retire_positions(m);
    return CHAD_OK;
}

chad_status chad_map_leaf_count(chad_map* m, int32_t submap, size_t* count) {
    if (!m || !count) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    return chad_map_export_leaves(m, submap, nullptr, nullptr, nullptr, 0, count);
}

chad_status chad_map_export_leaves(chad_map* m, int32_t submap, uint64_t* mc, float* sd, uint32_t* w, size_t capacity, size_t* count) {
    if (!m || !count) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    HostBricks hb;
    TRY(fetch_bricks(m, submap, hb));
    size_t k = 0;
    for (uint32_t b : hb.order) {
        for (uint32_t v = 0; v < 64; v++) {
            const uint32_t c = hb.cnt[size_t(b) * 64 + v];
            if (!c) continue;
            if (k < capacity) {
                if (mc) mc[k] = hb.keys[b] | v;
                if (sd) sd[k] = hb.sum[size_t(b) * 64 + v] / float(c);  // mean of the observations (octree.hpp:161-163)
                if (w) w[k] = c;
            }
            k++;
        }
    }
    *count = k;
    return CHAD_OK;
}

chad_status chad_map_stats(chad_map* m, chad_stats* out) {
    if (!m || !out) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    uint32_t overflow = 0, blocks = 0;
    TRY(read_back_header(m, &overflow));
    TRY(read_block_count(m, &blocks));
    collect_timing(m);
    std::memset(out, 0, sizeof(*out));
    const bool last_empty = m->frames_empty_tail != 0;   // the last insert had no points: its frame counters are zero
    out->frames = m->frames + m->frames_empty_tail; out->points_in = m->last_points_in; out->points_used = last_empty ? 0 : m->h_hdr->n_usable;
    out->updates = 0;
    if (!last_empty) for (uint32_t k = 0; k < kUpdateStripes; k++) out->updates += m->h_hdr->n_updates[k * 16];
    out->blocks = blocks; out->block_capacity = m->brick_capacity;
    out->hash_capacity = m->hash_capacity; out->closed_submaps = m->closed.size() + (m->closing ? 1 : 0);
    out->sort_passes = last_empty ? 0 : m->h_hdr->sort_passes; out->overflow = overflow;
    out->ms_mc_calc = m->ms[T_CALC]; out->ms_mc_sort = m->ms[T_SORT]; out->ms_norm_est = m->ms[T_NORM];
    out->ms_oct_upd = m->ms[T_UPD]; out->ms_total = m->ms[T_BEGIN];
    return CHAD_OK;
}

chad_status chad_map_host_times(chad_map* m, chad_host_times* out) {
    if (!m || !out) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    out->inserts = m->host_inserts; out->insert_s = m->host_insert_s; out->wait_s = m->host_wait_s; out->copy_s = m->host_copy_s;
    m->host_inserts = 0; m->host_insert_s = m->host_wait_s = m->host_copy_s = 0;
    return CHAD_OK;
}

chad_status chad_map_pipeline_times(chad_map* m, chad_pipeline_times* out) {
    if (!m || !out) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    if (out->struct_size < 2 * sizeof(uint32_t)) return fail(CHAD_ERR_INVALID_ARG, "chad_pipeline_times.struct_size not set");
    chad_pipeline_times t{};
    t.struct_size = uint32_t(std::min<size_t>(out->struct_size, sizeof(t)));
    t.rollovers = m->pt.rollovers; t.folds_completed = m->pt.folds_completed;
    t.level_growths = uint32_t(m->ddag.growths - m->growths_mark);
    t.rollover_host_s = m->pt.rollover_host_s; t.fold_host_s = m->pt.fold_host_s; t.fold_wait_s = m->pt.fold_wait_s;
    t.step_a_ms = m->pt.step_a_ms; t.step_b_ms = m->pt.step_b_ms;
    const AllocMeter& now = alloc_meter();
    t.mallocs = now.mallocs - m->alloc_mark.mallocs; t.frees = now.frees - m->alloc_mark.frees; t.malloc_bytes = now.bytes - m->alloc_mark.bytes;
    t.alloc_s = now.seconds - m->alloc_mark.seconds;
    t.dag_pool_bytes = m->ddag.pool.cap; t.dag_pool_used = m->ddag.pool.used; t.dag_pool_peak = m->ddag.pool.peak;
    t.exchanges = m->pt.exchanges; t.exchange_bytes = m->pt.exchange_bytes; t.exchange_union_blocks = m->pt.exchange_union_blocks;
    if (m->comm) { t.comm_world = uint32_t(m->comm->transport_world()); t.comm_transport = m->comm->nccl ? 1u : 2u; t.comm_p2p_calls = m->comm->p2p_calls; }
    std::memcpy(out, &t, t.struct_size);
    m->pt = chad_map::PipelineTimes{};
    m->alloc_mark = now; m->growths_mark = m->ddag.growths;
    return CHAD_OK;
}

chad_status chad_map_kernel_times(chad_map* m, chad_kernel_times* out) {
    if (!m || !out) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    if (!times_kernels(m)) return fail(CHAD_ERR_INVALID_ARG, "the map was not created with CHAD_FLAG_TIME_KERNELS");
    DeviceGuard guard(m->device);
    TRY(join_streams(m));
    collect_kernel_times(m);
    std::memset(out, 0, sizeof(*out));
    for (int id = 0; id < CHAD_KERNEL_COUNT; id++) {
        out->launches[id] = m->kernel_launches[id];
        out->ms[id] = m->kernel_launches[id] ? float(m->kernel_ms[id] / double(m->kernel_launches[id])) : 0.0f;
        m->kernel_ms[id] = 0; m->kernel_launches[id] = 0;
    }
    return CHAD_OK;
}

const char* chad_kernel_name(int32_t id) {
    static const char* const names[CHAD_KERNEL_COUNT] = {"k_frame_bounds", "k_frame_keys", "k_radix_onesweep[0]", "k_radix_onesweep[1]", "k_radix_onesweep[2]", "k_radix_onesweep[3]",
                                                         "k_radix_onesweep[4]", "k_radix_onesweep[5]", "k_radix_onesweep[6]", "k_radix_onesweep[7]", "k_frame_gather", "k_frame_normals", "k_update"};
    return id >= 0 && id < CHAD_KERNEL_COUNT ? names[id] : "";
}

chad_status chad_map_block_count(chad_map* m, int32_t submap, size_t* blocks) {
    if (!m || !blocks) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    uint32_t nb = 0;
    if (submap < 0) TRY(read_block_count(m, &nb));
    else {
        TRY(finish_closing(m, true));
        if (size_t(submap) >= m->closed.size()) return fail(CHAD_ERR_INVALID_ARG, "submap index out of range");
        nb = m->closed[size_t(submap)].blocks;
    }
    *blocks = nb;
    return CHAD_OK;
}

static chad_status export_blocks(chad_map* m, int32_t submap, uint64_t* keys_device, float* sum_device, uint32_t* cnt_device, size_t capacity_blocks, bool sorted) {
    if (!m) return fail(CHAD_ERR_INVALID_ARG, "null map");
    DeviceGuard guard(m->device);
    uint32_t nb = 0;
    if (submap < 0 && !sorted) {  // the multi-GPU merge's path: the live bricks straight into the caller's buffers, no intermediate copy
        TRY(read_block_count(m, &nb));
        if (nb > capacity_blocks) return fail(CHAD_ERR_INVALID_ARG, "capacity_blocks too small");
        if (!nb) return CHAD_OK;
        if (!keys_device || !sum_device || !cnt_device) return fail(CHAD_ERR_INVALID_ARG, "null output");
        HIP_TRY(hipMemsetAsync(m->d_cursor, 0, 4, m->stream));
        hipLaunchKernelGGL(k_compact_bricks, dim3(walk_grid(m->view.hash_mask)), dim3(256), 0, m->stream, m->view, m->d_cursor, nb, keys_device, sum_device, cnt_device, m->trunc);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(m->stream));
        return CHAD_OK;
    }
    const uint64_t* keys; const float* sum; const uint32_t* cnt;
    TRY(select_submap(m, submap, &nb, &keys, &sum, &cnt));
    if (nb > capacity_blocks) return fail(CHAD_ERR_INVALID_ARG, "capacity_blocks too small");
    if (!nb) return CHAD_OK;
    if (!keys_device || !sum_device || !cnt_device) return fail(CHAD_ERR_INVALID_ARG, "null output");
    if (!sorted) {
        HIP_TRY(hipMemcpyAsync(keys_device, keys, size_t(nb) * 8, hipMemcpyDeviceToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(sum_device, sum, size_t(nb) * 256, hipMemcpyDeviceToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(cnt_device, cnt, size_t(nb) * 256, hipMemcpyDeviceToDevice, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
        return CHAD_OK;
    }
    TRY(ensure_aux_sort(m, nb));
    // order the bricks by block key with the radix sorter (all 8 digits: keys are full 63-bit codes)
    RadixSortSignals sig;
    sig.fault = m->d_overflow;
    sig.dep = aux_sort_dep(m->aux_sort);   // (block keys: RadixSortSignals::dep)
    hipLaunchKernelGGL(k_copy_u64, grid_for(nb), dim3(256), 0, m->stream, keys, m->aux_sort.keys[0], nb);
    radix_sort_enqueue(m->aux_sort, nb, kMaxPasses, nullptr, m->stream, 0, 0, sig);
    hipLaunchKernelGGL(k_gather_bricks, grid_for(size_t(nb) * 64), dim3(256), 0, m->stream, (const uint32_t*)m->aux_sort.idx[0], nb, keys, sum, cnt, keys_device, sum_device, cnt_device);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(m->stream));
    return CHAD_OK;
}

chad_status chad_map_export_blocks_device(chad_map* m, int32_t submap, uint64_t* keys_device, float* sum_device, uint32_t* cnt_device, size_t capacity_blocks) {
    return export_blocks(m, submap, keys_device, sum_device, cnt_device, capacity_blocks, true);
}

chad_status chad_map_export_blocks_device_unordered(chad_map* m, int32_t submap, uint64_t* keys_device, float* sum_device, uint32_t* cnt_device, size_t capacity_blocks) {
    return export_blocks(m, submap, keys_device, sum_device, cnt_device, capacity_blocks, false);
}

chad_status chad_map_close_submap_with_blocks_device(chad_map* m, const uint64_t* keys_device, const float* sum_device, const uint32_t* cnt_device, size_t blocks) {
    if (!m || (blocks && (!keys_device || !sum_device || !cnt_device))) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    if (blocks > 0xffffffffull) return fail(CHAD_ERR_INVALID_ARG, "too many blocks");
    DeviceGuard guard(m->device);
    TRY(finish_closing(m, true));
    Submap s;
    TRY(alloc_submap(s, uint32_t(blocks)));
    if (blocks) {
        HIP_TRY(hipMemcpyAsync(s.keys, keys_device, blocks * 8, hipMemcpyDeviceToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(s.sum, sum_device, blocks * 256, hipMemcpyDeviceToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(s.cnt, cnt_device, blocks * 256, hipMemcpyDeviceToDevice, m->stream));
    }
    m->closed.push_back(s);
    retire_positions(m);
    TRY(clear_active(m));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return release_finalized(m);
}


// ---- multi-GPU: communicators (the exchange itself: merge_begin / merge_advance above) ---------------------------------------

chad_status chad_comm_unique_id(uint8_t id[CHAD_COMM_ID_BYTES]) {
    if (!id) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    const std::string err = rccl().load();
    if (!err.empty()) return fail(CHAD_ERR_HIP, err);
    static_assert(sizeof(RcclApi::UniqueId) == CHAD_COMM_ID_BYTES, "ncclUniqueId is 128 bytes (rccl.h:40)");
    RcclApi::UniqueId uid;
    const int r = rccl().get_unique_id(&uid);
    if (r) return fail(CHAD_ERR_HIP, std::string("ncclGetUniqueId: ") + rccl().get_error_string(r));
    std::memcpy(id, &uid, CHAD_COMM_ID_BYTES);
    return CHAD_OK;
}

chad_status chad_comm_create_rccl(const uint8_t id[CHAD_COMM_ID_BYTES], int32_t rank, int32_t world, chad_comm** out) {
    if (!out || !id) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return fail(CHAD_ERR_INVALID_ARG, "rank / world out of range (at most 15 ranks)");
    chad_comm* c = new chad_comm();
    c->rank = rank; c->world = world;
    c->no_shortcut = std::getenv("CHAD_DEBUG_COMM_NO_SHORTCUT") != nullptr;
    if (hipGetDevice(&c->device) != hipSuccess) { (void)hipGetLastError(); delete c; return fail(CHAD_ERR_NO_DEVICE, "no HIP device"); }
    if (c->exchanges()) {
        const std::string err = rccl().load();
        if (!err.empty()) { delete c; return fail(CHAD_ERR_HIP, err); }
        RcclApi::UniqueId uid;
        std::memcpy(&uid, id, CHAD_COMM_ID_BYTES);
        // ncclCommInitRank is collective: it returns once every rank has joined -- or never, when one does not.  It runs on a helper thread
        // that this call waits for with a limit (CHAD_COMM_TIMEOUT_S, default 120 s): a job whose ranks do not all arrive ends in an error
        // message here instead of a hang (the helper thread is then left behind; the process is expected to give up).
        struct Init { std::mutex mu; std::condition_variable cv; bool done = false; int result = 0; void* comm = nullptr; };
        auto st = std::make_shared<Init>();
        const int device = c->device;
        std::thread([st, uid, world, rank, device]() {
            (void)hipSetDevice(device);
            void* comm = nullptr;
            const int r = rccl().comm_init_rank(&comm, world, uid, rank);
            std::lock_guard<std::mutex> lock(st->mu);
            st->result = r; st->comm = comm; st->done = true;
            st->cv.notify_all();
        }).detach();
        {
            std::unique_lock<std::mutex> lock(st->mu);
            const bool arrived = comm_timeout_s() > 0 ? st->cv.wait_for(lock, std::chrono::duration<double>(comm_timeout_s()), [&] { return st->done; }) : (st->cv.wait(lock, [&] { return st->done; }), true);
            if (!arrived) { delete c; return fail(CHAD_ERR_HIP, "ncclCommInitRank did not return within " + std::to_string(int(comm_timeout_s())) + " s: not every rank of the job has joined (rank " + std::to_string(rank) + " of " + std::to_string(world) + ")"); }
            if (st->result) { delete c; return fail(CHAD_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().get_error_string(st->result)); }
            c->nccl = st->comm;
        }
        if (c->transport_world() != world) { const int seen = c->transport_world(); chad_comm_destroy(c); return fail(CHAD_ERR_HIP, "RCCL reports " + std::to_string(seen) + " ranks, the caller said " + std::to_string(world)); }
    }
    *out = c;
    return CHAD_OK;
}

chad_status chad_comm_create_callbacks(int32_t rank, int32_t world, chad_allgather_fn all_gather, chad_allreduce_fn all_reduce, void* user, chad_comm** out) {
    if (!out || (world > 1 && (!all_gather || !all_reduce))) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return fail(CHAD_ERR_INVALID_ARG, "rank / world out of range (at most 15 ranks)");
    chad_comm* c = new chad_comm();
    c->rank = rank; c->world = world; c->gather = all_gather; c->reduce = all_reduce; c->user = user;
    c->no_shortcut = std::getenv("CHAD_DEBUG_COMM_NO_SHORTCUT") != nullptr && all_gather && all_reduce;
    if (hipGetDevice(&c->device) != hipSuccess) { (void)hipGetLastError(); delete c; return fail(CHAD_ERR_NO_DEVICE, "no HIP device"); }
    *out = c;
    return CHAD_OK;
}

void chad_comm_destroy(chad_comm* c) {
    if (!c) return;
    if (c->nccl) (void)rccl().comm_destroy(c->nccl);
    delete c;
}

chad_status chad_map_attach_comm(chad_map* m, chad_comm* c) {
    if (!m) return fail(CHAD_ERR_INVALID_ARG, "null map");
    if (c && c->device != m->device) return fail(CHAD_ERR_INVALID_ARG, "the communicator was created on another device than the map's");
    if (c && (m->flags & CHAD_FLAG_SPACE_CARVING)) return fail(CHAD_ERR_INVALID_ARG, "space carving looks at the leaves already in the map: not defined for a map spread over ranks");
    if (m->closing_comm) {   // an exchange still in flight holds the old communicator (collective, like every other close point): finish it first
        DeviceGuard guard(m->device);
        TRY(finish_closing(m, true));
    }
    m->comm = c;
    m->stream_frames = 0;
    return CHAD_OK;
}

chad_status chad_map_merge_and_close(chad_map* m, chad_comm* c) {
    if (!m) return fail(CHAD_ERR_INVALID_ARG, "null map");
    DeviceGuard guard(m->device);
    TRY(merge_and_close(m, c ? c : m->comm));
    retire_positions(m);
    return finish_closing(m, true);   // an explicit exchange is complete when the call returns
}

// ---- Submap::finalize (submap.hpp:10-106): leaf clusters on the device, the DAG of NodeLevels on the host ----
namespace {

struct HostClusters { std::vector<uint64_t> keys, tsdf, weight; };

// Device half: the bricks of a dense submap, ordered by block key, as eight cluster words each -- left on the device in `out`
// (keys [nb], then tsdf [8 nb], then weight [8 nb] if wanted); `ms`: device time from here to `done` being recorded.
struct DeviceClusters {   // (in chad_map::cluster_arena: valid until the next call)
    uint64_t* keys = nullptr; uint64_t* tsdf = nullptr; uint64_t* weight = nullptr; uint8_t* mask = nullptr;   // mask[b]: which of block b's eight words hold leaves
};
chad_status device_cluster_words(chad_map* m, const uint64_t* keys, const float* sum, const uint32_t* cnt, uint32_t nb, bool want_weights, DeviceClusters& dc) {
    if (!nb) return CHAD_OK;
    TRY(ensure_aux_sort(m, nb));
    const size_t rows = Arena::padded(size_t(nb) * (8 + 64 + (want_weights ? 64 : 0))), need = rows + Arena::padded(nb);
    if (need > m->cluster_arena.cap) HIP_TRY(hipStreamSynchronize(m->stream));   // (the arena is about to move: nobody may still be reading the old one)
    HIP_TRY(m->cluster_arena.reserve(need));
    uint64_t* d_out = m->cluster_arena.take<uint64_t>(rows / 8 - 32);
    dc.keys = d_out; dc.tsdf = d_out + nb; dc.weight = want_weights ? dc.tsdf + size_t(nb) * 8 : nullptr;
    dc.mask = m->cluster_arena.take<uint8_t>(nb);
    // order the bricks by block key with the frame's radix sorter (all 8 digits: keys are full 63-bit codes)
    RadixSortSignals sig;
    sig.fault = m->d_overflow;
    sig.dep = aux_sort_dep(m->aux_sort);   // (block keys: RadixSortSignals::dep)
    hipLaunchKernelGGL(k_copy_u64, grid_for(nb), dim3(256), 0, m->stream, keys, m->aux_sort.keys[0], nb);
    radix_sort_enqueue(m->aux_sort, nb, kMaxPasses, nullptr, m->stream, 0, 0, sig);
    hipLaunchKernelGGL(k_leaf_clusters, grid_for(size_t(nb) * 64), dim3(256), 0, m->stream, (const uint32_t*)m->aux_sort.idx[0], nb, keys, sum, cnt, 1.0f / m->trunc /* submap.hpp:23 */,
                       dc.keys, dc.tsdf, dc.weight, dc.mask);
    HIP_TRY(hipGetLastError());
    return CHAD_OK;
}

struct TimedSpan {   // device time of a stretch of the map's stream (events released on every exit path)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~TimedSpan() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};

// ... and on the host (the host fold, chad_map_export_clusters)
chad_status device_clusters(chad_map* m, const uint64_t* keys, const float* sum, const uint32_t* cnt, uint32_t nb, bool want_weights, HostClusters& hc, float* ms) {
    hc.keys.resize(nb); hc.tsdf.resize(size_t(nb) * 8); hc.weight.resize(want_weights ? size_t(nb) * 8 : 0);
    if (!nb) return CHAD_OK;
    TimedSpan ev;
    HIP_TRY(hipEventCreate(&ev.e0)); HIP_TRY(hipEventCreate(&ev.e1));
    HIP_TRY(hipEventRecord(ev.e0, m->stream));
    DeviceClusters dc;
    TRY(device_cluster_words(m, keys, sum, cnt, nb, want_weights, dc));
    HIP_TRY(hipMemcpyAsync(hc.keys.data(), dc.keys, size_t(nb) * 8, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(hc.tsdf.data(), dc.tsdf, size_t(nb) * 64, hipMemcpyDeviceToHost, m->stream));
    if (want_weights) HIP_TRY(hipMemcpyAsync(hc.weight.data(), dc.weight, size_t(nb) * 64, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipEventRecord(ev.e1, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    float t = 0;
    (void)hipEventElapsedTime(&t, ev.e0, ev.e1);
    if (ms) *ms += t;
    return CHAD_OK;
}

// Fold the submaps closed since the last call into the DAG, in closing order (the order the reference finalizes them in).
// The fold in flight, one stage further if its numbers have arrived (block: wait for them).  When it completes, the submap's roots join
// dag_roots and the host's mirror of the level arrays is stale.
// A fold that fails half-way (out of memory for a level that has to grow, a launch error) is ABANDONED: its streams are drained, the
// fold state torn down and the DAG marked incomplete -- every chad_map_dag_* / iterator / finalize call reports that until
// chad_map_reset.  Without this the map re-entered the failed stage on every later call and could not even be reset.
static void fold_abort(chad_map* m) {
    if (m->fold_stream) (void)hipStreamSynchronize(m->fold_stream);
    if (m->close_stream) (void)hipStreamSynchronize(m->close_stream);
    (void)hipStreamSynchronize(m->stream);
    (void)hipGetLastError();
    m->folding = false; m->pending.set = false; m->fold.stage = 0;
    m->dag_broken = true;
}
static chad_status fold_step(chad_map* m, bool block);
chad_status fold_advance(chad_map* m, bool block) {
    if (m->dag_broken) { m->folding = false; m->pending.set = false; return CHAD_OK; }   // (nothing left to drive; the readers report it)
    const auto t0 = std::chrono::steady_clock::now();
    const chad_status st = fold_step(m, block);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (block ? m->pt.fold_wait_s : m->pt.fold_host_s) += dt;
    if (m->in_rollover_span) m->pt.rollover_host_s -= dt;   // (the counters are disjoint: a fold driven from inside a roll-over's bookkeeping counts as fold time)
    if (st != CHAD_OK) { const std::string why = g_error; fold_abort(m); g_error = "Submap::finalize on the device abandoned, the DAG is incomplete until chad_map_reset: " + why; }
    return st;
}
static chad_status fold_step(chad_map* m, bool block) {
    hipStream_t s = m->fold_stream ? m->fold_stream : m->stream;
    if (m->folding) {
        bool done = false;
        uint32_t root_tsdf = 0, root_weight = 0;
        const int stage_before = m->fold.stage;
        HIP_TRY(dag_fold_advance(m->ddag, m->fold, s, block, &done, &root_tsdf, &root_weight));
        if (m->fold.stage != stage_before && m->debug_fail_fold > 0 && --m->debug_fail_fold == 0) return fail(CHAD_ERR_OUT_OF_MEMORY, "CHAD_DEBUG_FAIL_FOLD: injected failure of a fold stage");
        if (m->fold.stage == 2 && stage_before == 1) HIP_TRY(hipEventRecord(m->fold_t1, s));   // (behind the last stage's kernels)
        if (!done) return CHAD_OK;
        float t = 0;
        (void)hipEventSynchronize(m->fold_t1);   // (recorded right behind the copy whose arrival ended the fold: a marker away)
        if (hipEventElapsedTime(&t, m->fold_t0, m->fold_t1) == hipSuccess) { m->dag_ms_device = t; m->pt.step_b_ms += t; } else (void)hipGetLastError();
        if (m->close_t0[m->fold_parity] && hipEventElapsedTime(&t, m->close_t0[m->fold_parity], m->close_t1[m->fold_parity]) == hipSuccess) m->pt.step_a_ms += t; else (void)hipGetLastError();
        m->pt.folds_completed++;
        m->dag_roots.push_back({root_tsdf, root_weight});
        m->dag_mirror_stale = true;
        m->folding = false;
    }
    if (m->pending.set) {   // the submap that was closed meanwhile: its fold begins behind its cluster words
        if (m->close_stream) HIP_TRY(hipStreamWaitEvent(s, m->clusters_ready, 0));
        HIP_TRY(hipEventRecord(m->fold_t0, s));
        HIP_TRY(dag_fold_begin(m->ddag, m->fold, m->pending.keys, m->pending.tsdf, m->pending.weight, m->pending.mask, m->pending.blocks, s));
        m->fold_parity = m->pending.parity;
        m->pending.set = false;
        m->folding = true;
    }
    return CHAD_OK;
}

// CHAD_FLAG_DAG_ONLY, second half of a roll-over: the closing table's block count is on the host.  Two steps, on a stream each, beside
// the frames and beside the map's stream:
//   A  (close_stream) the closing table's keys, their order, its bricks as cluster words in that order (k_table_keys, the sort,
//      k_sorted_clusters: no dense copy is ever made); the table is cleared behind that (map's stream) and is the spare again;
//   B  (fold_stream) the fold of those words into the DAG (dag_device.hpp), its second stage enqueued by the inserts that follow.
// B takes the submaps one after the other -- a level's addresses depend on everything added before -- but A needs nothing of the
// DAG: the A of a roll-over runs beside the B of the one before (the words in two arenas, taken in turn), and what paces a
// trajectory of cfg5's kind is the longer of the two steps instead of their sum.  Nothing here waits for the device unless TWO
// submaps are already on their way (one being folded, one waiting for that).
chad_status close_into_dag(chad_map* m, uint32_t blocks) {
    while (m->pending.set) TRY(fold_advance(m, true));   // (the fold before the waiting one is done: the waiting one begins; the arena of the one before it is free)
    if (!m->dag_broken && m->closed.size() > m->dag_roots.size() + (m->folding ? 1u : 0u)) TRY(fold_archived(m));   // submaps closed by other routes: the DAG takes them in closing order
    if (m->dag_broken) blocks = 0;   // (an earlier fold failed: the frames go on, the closed submaps are lost until chad_map_reset -- every DAG reader says so)
    hipStream_t s = m->close_stream ? m->close_stream : m->stream;
    const bool saturate = (m->flags & CHAD_FLAG_DAG_SATURATE_WEIGHTS) != 0;
    const size_t nb = std::max<uint32_t>(blocks, 1);
    const size_t words = nb * 8;
    const size_t need = Arena::padded(nb * 8) + Arena::padded(nb * 4) + (saturate ? 2 : 1) * Arena::padded(words * 8) + Arena::padded(nb);
    // (its last user was the fold of the roll-over before the one that may be in flight: done.  The count moves on only when this closing
    // is committed -- pending.set below: a step that fails in between leaves the parity where it was, so that the retry takes the SAME
    // arena and never the one the fold in flight is reading: ADVICE r3)
    const uint32_t parity = m->closings & 1u;
    Arena& arena = m->cluster_arenas[parity];
    if (need > arena.cap) {   // the arena is about to move: nothing may still be reading or writing the old one
        HIP_TRY(hipStreamSynchronize(s)); HIP_TRY(hipStreamSynchronize(m->stream));
        if (m->fold_stream) HIP_TRY(hipStreamSynchronize(m->fold_stream));
    }
    if (m->debug_fail_close > 0 && --m->debug_fail_close == 0) return fail(CHAD_ERR_OUT_OF_MEMORY, "CHAD_DEBUG_FAIL_CLOSE: injected failure of a roll-over's step A");
    HIP_TRY(arena.reserve(need));   // (grows with half as much again: the submaps of a trajectory differ by tens of per cent, and a hipMalloc stalls the device)
    uint64_t* keys = arena.take<uint64_t>(nb);
    uint32_t* slots = arena.take<uint32_t>(nb);
    uint64_t* tsdf = arena.take<uint64_t>(words);
    uint64_t* weight = saturate ? arena.take<uint64_t>(words) : nullptr;
    uint8_t* mask = arena.take<uint8_t>(nb);
    m->in_fold_setup = true;
    const chad_status sorter = blocks ? ensure_aux_sort(m, blocks) : CHAD_OK;
    m->in_fold_setup = false;
    TRY(sorter);
    if (s != m->stream) HIP_TRY(hipStreamWaitEvent(s, m->closing_counted, 0));   // behind the closing table's last update (the count was copied behind them)
    if (m->close_t0[parity]) HIP_TRY(hipEventRecord(m->close_t0[parity], s));
    if (blocks) {   // the live slots' keys, their order (all eight digits: full 63-bit codes), then the bricks in that order
        RadixSortSignals sig;
        sig.fault = m->d_overflow;
        sig.dep = aux_sort_dep(m->aux_sort);   // block keys: two of the eight passes usually have nothing to do
        HIP_TRY(hipMemsetAsync(m->d_cursor + 1, 0, 4, s));
        hipLaunchKernelGGL(k_table_keys, dim3(std::max(1u, std::min(2048u, ((m->closing_view.hash_mask >> 5) + 256u) / 256u))), dim3(256), 0, s, m->closing_view, m->d_cursor + 1, blocks, m->aux_sort.keys[0], slots);
        radix_sort_enqueue(m->aux_sort, blocks, kMaxPasses, nullptr, s, 0, 0, sig);
        HIP_TRY(hipMemcpyAsync(keys, m->aux_sort.keys[0], size_t(blocks) * 8, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_sorted_clusters, grid_for(size_t(blocks) * 16, 256, 4096), dim3(256), 0, s, m->closing_view, (const uint32_t*)m->aux_sort.idx[0], (const uint32_t*)slots, blocks, m->trunc,
                           1.0f / m->trunc /* submap.hpp:23 */, tsdf, weight, mask);
        HIP_TRY(hipGetLastError());
    }
    if (m->close_t1[parity]) HIP_TRY(hipEventRecord(m->close_t1[parity], s));
    // the table's clear: on the map's stream, behind the kernel that read the bricks; nothing the frames wait for is enqueued there
    // before the next roll-over
    if (s != m->stream) { HIP_TRY(hipEventRecord(m->clusters_ready, s)); HIP_TRY(hipStreamWaitEvent(m->stream, m->clusters_ready, 0)); }
    enqueue_clear(m->closing_view, m->stream);
    HIP_TRY(hipGetLastError());
    m->spare_flags = std::max(m->spare_flags, 1);
    m->spare = m->closing_view; m->has_spare = true;
    m->closing_view = MapView{}; m->closing = false;
    Submap gone;   // a closed submap without bricks: it is its DAG (read back with chad_map_dag_leaves, the iterator, the mesh from the DAG)
    gone.blocks = blocks; gone.released = true;
    m->closed.push_back(gone);
    m->pending.keys = keys; m->pending.tsdf = tsdf; m->pending.weight = weight; m->pending.mask = mask; m->pending.blocks = blocks; m->pending.parity = parity;
    m->pending.set = !m->dag_broken;
    m->closings++;
    m->pt.rollovers++;
    return fold_advance(m, false);   // (begins the fold at once if none is in flight)
}

// Fold the submaps closed since the last call into the DAG, in closing order (the order the reference finalizes them in).
chad_status finalize_closed(chad_map* m) {
    TRY(finish_closing(m, true));
    return fold_archived(m);
}
static chad_status dag_remote_error() { return fail(CHAD_ERR_INVALID_ARG, "this rank holds no DAG: the closed submaps of a frame-sharded map under CHAD_FLAG_DAG_ONLY are folded on rank 0 (read them back there)"); }
static chad_status dag_broken_error() { return fail(CHAD_ERR_HIP, "the DAG is incomplete: a fold into it failed earlier (chad_last_error at that call said why); chad_map_reset starts a new one"); }
// ... those that were archived as dense copies (the default; the exchange's result; a closing without a spare table), start to finish
chad_status fold_archived(chad_map* m) {
    while (fold_busy(m)) TRY(fold_advance(m, true));
    if (m->dag_broken) return dag_broken_error();
    if (m->dag_roots.size() >= m->closed.size()) return CHAD_OK;
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->fold_stream) HIP_TRY(hipStreamSynchronize(m->fold_stream));   // (the folds below run on the map's stream: behind whatever a level's growth left on the other one)
    const bool saturate = (m->flags & CHAD_FLAG_DAG_SATURATE_WEIGHTS) != 0;
    float ms_device = 0;
    double ms_host = 0;
    while (m->dag_roots.size() < m->closed.size()) {
        const Submap& s = m->closed[m->dag_roots.size()];
        uint32_t root_tsdf = 0, root_weight = 0;
        if (m->dag_on_host) {   // CHAD_DEBUG_DAG_HOST: cluster words to the host, the sequential fold of dag.hpp
            HostClusters hc;
            TRY(device_clusters(m, s.keys, s.sum, s.cnt, s.blocks, saturate, hc, &ms_device));
            const auto t0 = std::chrono::steady_clock::now();
            dag_add_submap(m->dag, hc.keys.data(), hc.tsdf.data(), saturate ? hc.weight.data() : nullptr, s.blocks, &root_tsdf, &root_weight);
            ms_host += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        } else {                // the DAG lives on the device: level by level, bottom-up (dag_device.hpp)
            TimedSpan ev;
            HIP_TRY(hipEventCreate(&ev.e0)); HIP_TRY(hipEventCreate(&ev.e1));
            HIP_TRY(hipEventRecord(ev.e0, m->stream));
            DeviceClusters dc;
            TRY(device_cluster_words(m, s.keys, s.sum, s.cnt, s.blocks, saturate, dc));
            const hipError_t folded = dag_device_add_submap(m->ddag, m->fold, dc.keys, dc.tsdf, saturate ? dc.weight : nullptr, dc.mask, s.blocks, m->stream, &root_tsdf, &root_weight);
            if (folded != hipSuccess) fold_abort(m);   // (half a submap's nodes may be in the levels: the DAG is incomplete from here on)
            HIP_TRY(folded);
            HIP_TRY(hipEventRecord(ev.e1, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
            float t = 0;
            (void)hipEventElapsedTime(&t, ev.e0, ev.e1);
            ms_device += t;
            m->dag_mirror_stale = true;
        }
        m->dag_roots.push_back({root_tsdf, root_weight});
    }
    m->dag_ms_device = ms_device; m->dag_ms_host = float(ms_host);
    return CHAD_OK;
}

// The host's copy of the level arrays, up to date (the device fold leaves the new words on the device until somebody reads them here).
chad_status dag_host(chad_map* m) {
    if (!m->dag_mirror_stale) return CHAD_OK;
    const auto t0 = std::chrono::steady_clock::now();
    if (m->fold_stream) HIP_TRY(hipStreamSynchronize(m->fold_stream));   // (a level that grew behind the last fold is being copied there)
    HIP_TRY(dag_device_mirror(m->ddag, m->dag, m->stream));
    m->dag_ms_host += float(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    m->dag_mirror_stale = false;
    return CHAD_OK;
}

}  // namespace

chad_status chad_map_finalize(chad_map* m) {
    if (!m) return fail(CHAD_ERR_INVALID_ARG, "null map");
    DeviceGuard guard(m->device);
    return finalize_closed(m);
}

chad_status chad_map_dag_roots(chad_map* m, int32_t submap, uint32_t* root_tsdf, uint32_t* root_weight) {
    if (!m || !root_tsdf || !root_weight) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    TRY(finalize_closed(m));
    if (m->dag_remote) return dag_remote_error();
    if (submap < 0 || size_t(submap) >= m->dag_roots.size()) return fail(CHAD_ERR_INVALID_ARG, "submap index out of range (only closed submaps are in the DAG)");
    *root_tsdf = m->dag_roots[size_t(submap)][0]; *root_weight = m->dag_roots[size_t(submap)][1];
    return CHAD_OK;
}

chad_status chad_map_dag_level_words(chad_map* m, uint32_t level, size_t* words) {
    if (!m || !words) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    if (level > CHAD_DAG_CLUSTER_LEVEL) return fail(CHAD_ERR_INVALID_ARG, "level out of range");
    DeviceGuard guard(m->device);
    TRY(finalize_closed(m));
    if (m->dag_remote) return dag_remote_error();
    if (!m->dag_on_host && m->ddag.ready) { *words = m->ddag.h_scalars[level].size; return CHAD_OK; }   // (no need to bring the words over to count them)
    *words = level < kDagDepth ? m->dag.nodes[level].raw.size() : m->dag.clusters.raw.size();
    return CHAD_OK;
}

chad_status chad_map_dag_level_copy(chad_map* m, uint32_t level, void* out, size_t capacity_words) {
    size_t words = 0;
    TRY(chad_map_dag_level_words(m, level, &words));
    if (capacity_words < words) return fail(CHAD_ERR_INVALID_ARG, "capacity_words too small");
    if (!out) return fail(CHAD_ERR_INVALID_ARG, "null output");
    DeviceGuard guard(m->device);
    TRY(dag_host(m));
    if (level < kDagDepth) std::memcpy(out, m->dag.nodes[level].raw.data(), words * 4);
    else std::memcpy(out, m->dag.clusters.raw.data(), words * 8);
    return CHAD_OK;
}

chad_status chad_map_dag_reserve(chad_map* m, const uint64_t entries[CHAD_DAG_NODE_LEVELS + 1], const uint64_t words[CHAD_DAG_NODE_LEVELS + 1]) {
    if (!m || !entries || !words) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    for (uint32_t l = 0; l <= kDagDepth; l++) {
        if (entries[l] >= (1ull << 30) || words[l] >= (1ull << 32)) return fail(CHAD_ERR_INVALID_ARG, "a level of more than 2^30 values or 2^32 words");
        m->ddag.hint_entries[l] = entries[l]; m->ddag.hint_words[l] = words[l];
    }
    return CHAD_OK;   // (takes effect when the DAG's stores are made: the first fold of a map, or the first after chad_map_reset)
}

chad_status chad_map_dag_stats(chad_map* m, chad_dag_stats* out) {
    if (!m || !out) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    if (out->struct_size < 2 * sizeof(uint32_t)) return fail(CHAD_ERR_INVALID_ARG, "chad_dag_stats.struct_size not set");
    TRY(finalize_closed(m));
    chad_dag_stats filled{};
    filled.struct_size = uint32_t(std::min<size_t>(out->struct_size, sizeof(filled)));
    chad_dag_stats* const caller = out;
    out = &filled;
    out->submaps = m->dag_roots.size();
    if (!m->dag_on_host && m->ddag.ready)
        for (uint32_t l = 0; l <= kDagDepth; l++) { out->uniques[l] = uint32_t(m->ddag.h_scalars[l].uniques); out->dupes[l] = uint32_t(m->ddag.h_scalars[l].dupes); }
    else {
        for (uint32_t l = 0; l < kDagDepth; l++) { out->uniques[l] = m->dag.nodes[l].uniques; out->dupes[l] = m->dag.nodes[l].dupes; }
        out->uniques[kDagDepth] = m->dag.clusters.uniques; out->dupes[kDagDepth] = m->dag.clusters.dupes;
    }
    out->ms_device = m->dag_ms_device; out->ms_host = m->dag_ms_host;
    out->device_bytes = m->ddag.arena_cap + m->ddag.outgrown_bytes + m->ddag.pool.cap;   // (the pool's whole reservation, used or not; beside it what did not fit)
    for (uint32_t l = 0; l <= kDagDepth && m->ddag.ready; l++) {
        const DagLevelDev& L = m->ddag.level[l];
        if (!m->ddag.pool.owns(L.table)) out->device_bytes += size_t(L.table_slots) * 4;
        if (!m->ddag.pool.owns(L.raw)) out->device_bytes += size_t(L.raw_capacity) * (l == kDagDepth ? 8 : 4);
    }
    std::memcpy(caller, &filled, filled.struct_size);   // (an older, shorter caller gets the fields it knows)
    return CHAD_OK;
}

chad_status chad_map_dag_leaves(chad_map* m, int32_t submap, uint64_t* mc, float* sd, size_t capacity, size_t* count) {
    if (!m || !count) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    TRY(finalize_closed(m));
    if (m->dag_remote) return dag_remote_error();
    if (submap < 0 || size_t(submap) >= m->dag_roots.size()) return fail(CHAD_ERR_INVALID_ARG, "submap index out of range (only closed submaps are in the DAG)");
    TRY(dag_host(m));
    size_t k = 0;
    const float trunc = m->trunc;
    dag_walk(m->dag, m->dag_roots[size_t(submap)][0], [&](uint64_t code, uint8_t bits) {
        if (k < capacity) {
            if (mc) mc[k] = code;
            if (sd) sd[k] = cluster_sd_of_bits(bits, trunc);
        }
        k++;
    });
    *count = k;
    return CHAD_OK;
}

chad_status chad_map_export_clusters(chad_map* m, int32_t submap, uint64_t* keys_out, uint64_t* tsdf_out, uint64_t* weights_out, size_t capacity_blocks, size_t* blocks) {
    if (!m || !blocks) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    uint32_t nb = 0;
    const uint64_t* keys; const float* sum; const uint32_t* cnt;
    TRY(select_submap(m, submap, &nb, &keys, &sum, &cnt));
    *blocks = nb;
    if (!keys_out && !tsdf_out && !weights_out) return CHAD_OK;
    if (nb > capacity_blocks) return fail(CHAD_ERR_INVALID_ARG, "capacity_blocks too small");
    HostClusters hc;
    TRY(device_clusters(m, keys, sum, cnt, nb, weights_out != nullptr, hc, nullptr));
    if (keys_out) std::memcpy(keys_out, hc.keys.data(), size_t(nb) * 8);
    if (tsdf_out) std::memcpy(tsdf_out, hc.tsdf.data(), size_t(nb) * 64);
    if (weights_out) std::memcpy(weights_out, hc.weight.data(), size_t(nb) * 64);
    return CHAD_OK;
}


// ---- read-back: leaf iterator over a closed submap's DAG (tsdf.hpp:120-161, tsdf.cpp:88-159) -----------------------------------
struct chad_iter {
    chad_map* map;
    uint64_t generation;
    int32_t submap;
    DagCursor cursor;
};

// p <- R p + t of closed submap `submap`, if it has been given a pose
static void apply_submap_pose(const chad_map* m, int32_t submap, float p[3]) {
    if (submap < 0 || size_t(submap) >= m->closed_pose.size()) return;
    const std::array<float, 12>& T = m->closed_pose[size_t(submap)];
    const float x = p[0], y = p[1], z = p[2];
    p[0] = T[0] * x + T[1] * y + T[2] * z + T[9];
    p[1] = T[3] * x + T[4] * y + T[5] * z + T[10];
    p[2] = T[6] * x + T[7] * y + T[8] * z + T[11];
}

// n <- R n (a direction: no translation)
static void apply_submap_rotation(const chad_map* m, int32_t submap, float n[3]) {
    if (submap < 0 || size_t(submap) >= m->closed_pose.size()) return;
    const std::array<float, 12>& T = m->closed_pose[size_t(submap)];
    const float x = n[0], y = n[1], z = n[2];
    n[0] = T[0] * x + T[1] * y + T[2] * z; n[1] = T[3] * x + T[4] * y + T[5] * z; n[2] = T[6] * x + T[7] * y + T[8] * z;
}

chad_status chad_map_iter_begin(chad_map* m, int32_t submap, chad_iter** out) {
    if (!m || !out) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    DeviceGuard guard(m->device);
    TRY(finalize_closed(m));
    if (m->dag_remote) return dag_remote_error();
    // tsdf.cpp:106-110: "chad::TSDFMap::begin() given faulty root address"
    if (submap < 0 || size_t(submap) >= m->dag_roots.size()) return fail(CHAD_ERR_INVALID_ARG, "submap index out of range (only closed submaps are in the DAG)");
    TRY(dag_host(m));
    chad_iter* it = new chad_iter{m, m->dag_generation, submap, DagCursor{}};
    it->cursor.open(m->dag, m->dag_roots[size_t(submap)][0], m->dag_roots[size_t(submap)][1]);
    it->cursor.first();
    m->iterators.push_back(it);
    *out = it;
    return CHAD_OK;
}

void chad_iter_destroy(chad_iter* it) {
    if (!it) return;
    if (it->map) {
        auto& list = it->map->iterators;
        list.erase(std::remove(list.begin(), list.end(), it), list.end());
    }
    delete it;
}

static void orphan_iterators(chad_map* m) {
    for (chad_iter* it : m->iterators) it->map = nullptr;
    m->iterators.clear();
}
static bool iter_alive(const chad_iter* it) { return it && it->map && it->generation == it->map->dag_generation; }

int32_t chad_iter_valid(const chad_iter* it) { return iter_alive(it) && it->cursor.valid ? 1 : 0; }

chad_status chad_iter_seek(chad_iter* it, int32_t to_last) {
    if (!iter_alive(it)) return fail(CHAD_ERR_INVALID_ARG, "dead iterator (the map was reset)");
    if (to_last) it->cursor.last(); else it->cursor.first();
    return CHAD_OK;
}

chad_status chad_iter_advance(chad_iter* it, int64_t n, int64_t* moved) {
    if (!iter_alive(it)) return fail(CHAD_ERR_INVALID_ARG, "dead iterator (the map was reset)");
    int64_t done = 0;
    for (; done < (n < 0 ? -n : n) && it->cursor.valid; done++)
        if (!(n < 0 ? it->cursor.prev() : it->cursor.next())) break;
    if (moved) *moved = n < 0 ? -done : done;
    return CHAD_OK;
}

static void leaf_of(const chad_iter* it, chad_leaf* out) {
    const chad_map* m = it->map;
    const uint64_t code = it->cursor.code();
    // the leaf's position: its voxel's corner, vec3(decode(mc)) * res (octree.hpp:157, lvr2.cpp:78-79)
    float p[3] = {float(int(dag_compact21(code) - (1u << 20))) * m->res, float(int(dag_compact21(code >> 1) - (1u << 20))) * m->res, float(int(dag_compact21(code >> 2) - (1u << 20))) * m->res};
    apply_submap_pose(m, it->submap, p);   // identity unless chad_map_set_submap_pose has re-posed the submap (loop closure)
    out->x = p[0]; out->y = p[1]; out->z = p[2];
    out->signed_distance = cluster_sd_of_bits(it->cursor.sd_bits(), m->trunc);
    out->weight = it->cursor.weight_bits();
    out->reserved = 0;
}

chad_status chad_iter_get(const chad_iter* it, chad_leaf* out) {
    if (!iter_alive(it) || !out) return fail(CHAD_ERR_INVALID_ARG, "dead iterator or null output");
    if (!it->cursor.valid) return fail(CHAD_ERR_INVALID_ARG, "iterator is past the end");
    leaf_of(it, out);
    return CHAD_OK;
}

chad_status chad_iter_read(chad_iter* it, chad_leaf* out, size_t capacity, size_t* count) {
    if (!iter_alive(it) || !count || (capacity && !out)) return fail(CHAD_ERR_INVALID_ARG, "dead iterator or null argument");
    size_t k = 0;
    while (k < capacity && it->cursor.valid) {
        leaf_of(it, &out[k++]);
        it->cursor.next();
    }
    *count = k;
    return CHAD_OK;
}

// ---- loop-closure support (EXTENSION; upstream roadmap README.md:59): a submap's frame poses out, a rigid correction in ---------
chad_status chad_map_submap_positions(chad_map* m, int32_t submap, float* xyz, size_t capacity, size_t* count) {
    if (!m || !count) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    const std::vector<std::array<float, 3>>* poses = &m->positions;
    if (submap >= 0) {
        if (size_t(submap) >= m->closed_positions.size()) return fail(CHAD_ERR_INVALID_ARG, "submap index out of range");
        poses = &m->closed_positions[size_t(submap)];
    }
    *count = poses->size();
    if (xyz) for (size_t k = 0; k < std::min(capacity, poses->size()); k++) std::memcpy(xyz + 3 * k, (*poses)[k].data(), 12);
    return CHAD_OK;
}

chad_status chad_map_set_submap_pose(chad_map* m, int32_t submap, const float rotation[9], const float translation[3]) {
    if (!m || !rotation || !translation) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    if (submap < 0 || size_t(submap) >= m->closed_positions.size()) return fail(CHAD_ERR_INVALID_ARG, "only a closed submap can be re-posed");
    if (m->closed_pose.size() <= size_t(submap)) m->closed_pose.resize(size_t(submap) + 1, std::array<float, 12>{1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0});
    std::memcpy(m->closed_pose[size_t(submap)].data(), rotation, 36);
    std::memcpy(m->closed_pose[size_t(submap)].data() + 9, translation, 12);
    return CHAD_OK;
}

// ---- read-back: batched queries of the ACTIVE submap on the device (query_kernels.hpp) ---------------------------------------
static chad_status query_ready(chad_map* m) {
    TRY(flush_pending_clear(m));
    return join_streams(m);   // the frames enqueued so far are part of what is asked about
}

chad_status chad_map_query_sd_device(chad_map* m, const float* xyz_device, size_t n, int32_t mode, float* sd_device, uint32_t* weight_device) {
    if (!m || (n && (!xyz_device || !sd_device))) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    if (mode != CHAD_QUERY_LEAF && mode != CHAD_QUERY_TRILINEAR) return fail(CHAD_ERR_INVALID_ARG, "unknown query mode");
    if (n >= (1ull << 32)) return fail(CHAD_ERR_INVALID_ARG, "too many query points");
    DeviceGuard guard(m->device);
    if (!n) return CHAD_OK;
    TRY(query_ready(m));
    hipLaunchKernelGGL(k_query_points, grid_for(n, 256, 4096), dim3(256), 0, m->stream, m->view, xyz_device, uint32_t(n), uint32_t(mode), m->res, m->recip, m->trunc, sd_device, weight_device);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(m->stream));
    return CHAD_OK;
}

chad_status chad_map_query_sd(chad_map* m, const float* xyz, size_t n, int32_t mode, float* sd, uint32_t* weight) {
    if (!m || (n && (!xyz || !sd))) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    if (!n) return CHAD_OK;
    DevBuf d_xyz, d_sd, d_w;
    HIP_TRY(d_xyz.alloc(n * 12)); HIP_TRY(d_sd.alloc(n * 4)); HIP_TRY(d_w.alloc(n * 4));
    HIP_TRY(hipMemcpy(d_xyz.p, xyz, n * 12, hipMemcpyHostToDevice));
    TRY(chad_map_query_sd_device(m, d_xyz.as<float>(), n, mode, d_sd.as<float>(), d_w.as<uint32_t>()));
    HIP_TRY(hipMemcpy(sd, d_sd.p, n * 4, hipMemcpyDeviceToHost));
    if (weight) HIP_TRY(hipMemcpy(weight, d_w.p, n * 4, hipMemcpyDeviceToHost));
    return CHAD_OK;
}

chad_status chad_map_raycast(chad_map* m, const float* origins, const float* directions, size_t n, float t_max, float* t_hit) {
    if (!m || (n && (!origins || !directions || !t_hit))) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    if (n >= (1ull << 32) || !(t_max >= 0.0f)) return fail(CHAD_ERR_INVALID_ARG, "too many rays or negative range");
    DeviceGuard guard(m->device);
    if (!n) return CHAD_OK;
    DevBuf d_o, d_d, d_t;
    HIP_TRY(d_o.alloc(n * 12)); HIP_TRY(d_d.alloc(n * 12)); HIP_TRY(d_t.alloc(n * 4));
    HIP_TRY(hipMemcpy(d_o.p, origins, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_d.p, directions, n * 12, hipMemcpyHostToDevice));
    TRY(query_ready(m));
    hipLaunchKernelGGL(k_raycast, grid_for(n, 256, 4096), dim3(256), 0, m->stream, m->view, d_o.as<float>(), d_d.as<float>(), uint32_t(n), t_max, m->res, m->recip, m->trunc, d_t.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(t_hit, d_t.p, n * 4, hipMemcpyDeviceToHost));
    return CHAD_OK;
}

// ---- mesh extraction (reference: TSDFMap::save -> lvr2.cpp:275-321) -----------------------------------
namespace {

// triangle soup -> indexed mesh: vertices welded by their bits (mesh_kernels.hpp: a grid edge's vertex comes out bit-identical in
// every cell around it), triangles that lost a corner to the welding (a surface passing exactly through a grid point) dropped
void weld_mesh(const std::vector<float>& soup, std::vector<float>& vertices, std::vector<uint32_t>& triangles) {
    struct Key { uint32_t x, y, z; bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; } };
    struct Hash { size_t operator()(const Key& k) const { return size_t(dag_mix((uint64_t(k.x) << 32 | k.y) ^ dag_mix(k.z))); } };
    std::unordered_map<Key, uint32_t, Hash> index;
    index.reserve(soup.size() / 9 + 16);
    vertices.clear(); triangles.clear();
    for (size_t t = 0; t + 9 <= soup.size(); t += 9) {
        uint32_t v[3];
        for (int c = 0; c < 3; c++) {
            Key k;
            float p[3] = {soup[t + 3 * c], soup[t + 3 * c + 1], soup[t + 3 * c + 2]};
            for (float& f : p) if (f == 0.0f) f = 0.0f;   // -0 and +0 are the same place
            std::memcpy(&k.x, &p[0], 4); std::memcpy(&k.y, &p[1], 4); std::memcpy(&k.z, &p[2], 4);
            auto found = index.find(k);
            if (found == index.end()) {
                found = index.emplace(k, uint32_t(vertices.size() / 3)).first;
                vertices.insert(vertices.end(), p, p + 3);
            }
            v[c] = found->second;
        }
        if (v[0] == v[1] || v[1] == v[2] || v[0] == v[2]) continue;
        triangles.insert(triangles.end(), v, v + 3);
    }
}

// What the reference saves with its mesh (lvr2.cpp:296-313: calcFaceNormals, calcVertexNormals, SimpleFinalizer::setNormalData): a normal
// per vertex -- the normalised sum of the unit normals of the faces around it (a face without area contributes nothing; a vertex
// without faces gets (0, 0, 0)).
void mesh_vertex_normals(const std::vector<float>& vertices, const std::vector<uint32_t>& triangles, std::vector<float>& normals) {
    normals.assign(vertices.size(), 0.0f);
    for (size_t t = 0; t + 3 <= triangles.size(); t += 3) {
        const float* a = &vertices[3 * size_t(triangles[t])], *b = &vertices[3 * size_t(triangles[t + 1])], *c = &vertices[3 * size_t(triangles[t + 2])];
        const double ux = double(b[0]) - a[0], uy = double(b[1]) - a[1], uz = double(b[2]) - a[2], vx = double(c[0]) - a[0], vy = double(c[1]) - a[1], vz = double(c[2]) - a[2];
        double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
        const double len = std::sqrt(nx * nx + ny * ny + nz * nz);
        if (!(len > 0.0)) continue;
        nx /= len; ny /= len; nz /= len;
        for (int k = 0; k < 3; k++) { float* n = &normals[3 * size_t(triangles[t + k])]; n[0] += float(nx); n[1] += float(ny); n[2] += float(nz); }
    }
    for (size_t v = 0; v + 3 <= normals.size(); v += 3) {
        const double len = std::sqrt(double(normals[v]) * normals[v] + double(normals[v + 1]) * normals[v + 1] + double(normals[v + 2]) * normals[v + 2]);
        if (len > 0.0) { normals[v] = float(normals[v] / len); normals[v + 1] = float(normals[v + 1] / len); normals[v + 2] = float(normals[v + 2] / len); }
    }
}

// sorted block keys and per-voxel means (NaN = no leaf) of a submap on the device
chad_status mesh_source(chad_map* m, int32_t submap, DevBuf& keys_sorted, DevBuf& means, uint32_t* blocks_out) {
    TRY(finish_closing(m, true));   // (a roll-over still in progress: its archive may be released right behind it, CHAD_FLAG_DAG_ONLY)
    const bool from_dag = submap >= 0 && ((m->flags & CHAD_FLAG_MESH_FROM_DAG) || (size_t(submap) < m->closed.size() && m->closed[size_t(submap)].released));
    if (from_dag) {
        // what the reference meshes (lvr2.cpp:30-90): the leaves read back through the DAG, signed distances de-quantised from 8 bits
        TRY(finalize_closed(m));
        if (m->dag_remote) return dag_remote_error();
        if (size_t(submap) >= m->dag_roots.size()) return fail(CHAD_ERR_INVALID_ARG, "submap index out of range");
        TRY(dag_host(m));
        std::vector<uint64_t> keys;
        std::vector<float> sd;
        DagCursor c;
        c.open(m->dag, m->dag_roots[size_t(submap)][0], m->dag_roots[size_t(submap)][1]);
        for (bool ok = c.first(); ok; ok = c.next()) {
            const uint64_t code = c.code();
            if (keys.empty() || keys.back() != (code & kBlockMask)) { keys.push_back(code & kBlockMask); sd.resize(sd.size() + 64, std::nanf("")); }
            sd[sd.size() - 64 + (code & 63u)] = cluster_sd_of_bits(c.sd_bits(), m->trunc);
        }
        *blocks_out = uint32_t(keys.size());
        HIP_TRY(keys_sorted.alloc(keys.size() * 8)); HIP_TRY(means.alloc(sd.size() * 4));
        if (!keys.empty()) {
            HIP_TRY(hipMemcpy(keys_sorted.p, keys.data(), keys.size() * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(means.p, sd.data(), sd.size() * 4, hipMemcpyHostToDevice));
        }
        return CHAD_OK;
    }
    uint32_t nb = 0;
    const uint64_t* keys; const float* sum; const uint32_t* cnt;
    TRY(select_submap(m, submap, &nb, &keys, &sum, &cnt));
    *blocks_out = nb;
    HIP_TRY(keys_sorted.alloc(size_t(nb) * 8)); HIP_TRY(means.alloc(size_t(nb) * 256));
    if (!nb) return CHAD_OK;
    TRY(ensure_aux_sort(m, nb));
    RadixSortSignals sig;
    sig.fault = m->d_overflow;
    sig.dep = aux_sort_dep(m->aux_sort);   // (block keys: RadixSortSignals::dep)
    hipLaunchKernelGGL(k_copy_u64, grid_for(nb), dim3(256), 0, m->stream, keys, m->aux_sort.keys[0], nb);
    radix_sort_enqueue(m->aux_sort, nb, kMaxPasses, nullptr, m->stream, 0, 0, sig);
    hipLaunchKernelGGL(k_brick_means, grid_for(size_t(nb) * 64), dim3(256), 0, m->stream, (const uint32_t*)m->aux_sort.idx[0], nb, keys, sum, cnt, keys_sorted.as<uint64_t>(), means.as<float>());
    HIP_TRY(hipGetLastError());
    return CHAD_OK;
}

chad_status extract_mesh(chad_map* m, int32_t submap) {
    if (m->mesh_submap == submap && submap >= 0) return CHAD_OK;   // (a closed submap does not change; the active one does)
    DevBuf keys, means;
    uint32_t nb = 0;
    TRY(mesh_source(m, submap, keys, means, &nb));
    m->mesh_vertices.clear(); m->mesh_triangles.clear(); m->mesh_normals.clear();
    m->mesh_submap = submap;
    if (!nb) return CHAD_OK;
    const bool cubes = !(m->flags & CHAD_FLAG_MESH_TETRAHEDRA);
    DevBuf counts, d_total;
    HIP_TRY(counts.alloc(size_t(nb) * 4)); HIP_TRY(d_total.alloc(4));
    auto bricks = [&](auto write, float* out) {
        constexpr bool kWrite = decltype(write)::value;
        if (cubes) hipLaunchKernelGGL((k_mesh_bricks<kWrite, true>), grid_for(size_t(nb) * 64, 256, 8192), dim3(256), 0, m->stream, keys.as<uint64_t>(), means.as<float>(), nb, m->res, counts.as<uint32_t>(), out);
        else hipLaunchKernelGGL((k_mesh_bricks<kWrite, false>), grid_for(size_t(nb) * 64, 256, 8192), dim3(256), 0, m->stream, keys.as<uint64_t>(), means.as<float>(), nb, m->res, counts.as<uint32_t>(), out);
    };
    bricks(std::false_type{}, nullptr);
    hipLaunchKernelGGL(k_unique_offsets, dim3(1), dim3(256), 0, m->stream, counts.as<uint32_t>(), nb, d_total.as<uint32_t>());
    uint32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_total.p, 4, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (!total) return CHAD_OK;
    DevBuf soup;
    HIP_TRY(soup.alloc(size_t(total) * 36));
    bricks(std::true_type{}, soup.as<float>());
    HIP_TRY(hipGetLastError());
    std::vector<float> host(size_t(total) * 9);
    HIP_TRY(hipMemcpyAsync(host.data(), soup.p, host.size() * 4, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    weld_mesh(host, m->mesh_vertices, m->mesh_triangles);
    mesh_vertex_normals(m->mesh_vertices, m->mesh_triangles, m->mesh_normals);
    return CHAD_OK;
}

}  // namespace

chad_status chad_map_mesh(chad_map* m, int32_t submap, float* vertices, size_t vertex_capacity, uint32_t* triangles, size_t triangle_capacity, size_t* vertex_count, size_t* triangle_count) {
    if (!m || !vertex_count || !triangle_count) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    TRY(extract_mesh(m, submap));
    *vertex_count = m->mesh_vertices.size() / 3; *triangle_count = m->mesh_triangles.size() / 3;
    if (vertices && vertex_capacity >= *vertex_count) {
        std::memcpy(vertices, m->mesh_vertices.data(), m->mesh_vertices.size() * 4);
        for (size_t v = 0; v < *vertex_count; v++) apply_submap_pose(m, submap, vertices + 3 * v);   // (loop closure: the submap's corrected pose)
    }
    if (triangles && triangle_capacity >= *triangle_count) std::memcpy(triangles, m->mesh_triangles.data(), m->mesh_triangles.size() * 4);
    if ((vertices && vertex_capacity < *vertex_count) || (triangles && triangle_capacity < *triangle_count)) return fail(CHAD_ERR_INVALID_ARG, "mesh buffers too small");
    return CHAD_OK;
}

chad_status chad_map_mesh_normals(chad_map* m, int32_t submap, float* normals, size_t vertex_capacity, size_t* vertex_count) {
    if (!m || !vertex_count) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    TRY(extract_mesh(m, submap));
    *vertex_count = m->mesh_normals.size() / 3;
    if (!normals) return CHAD_OK;
    if (vertex_capacity < *vertex_count) return fail(CHAD_ERR_INVALID_ARG, "mesh buffers too small");
    std::memcpy(normals, m->mesh_normals.data(), m->mesh_normals.size() * 4);
    for (size_t v = 0; v < *vertex_count; v++) apply_submap_rotation(m, submap, normals + 3 * v);   // (loop closure: the submap's corrected pose turns the normals with it)
    return CHAD_OK;
}

chad_status chad_map_mesh_cells(chad_map* m, int32_t submap, uint64_t* codes, float* corners, size_t capacity, size_t* count) {
    if (!m || !count) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    DevBuf keys, means;
    uint32_t nb = 0;
    TRY(mesh_source(m, submap, keys, means, &nb));
    *count = 0;
    if (!nb) return CHAD_OK;
    DevBuf counts, d_total;
    HIP_TRY(counts.alloc(size_t(nb) * 4)); HIP_TRY(d_total.alloc(4));
    hipLaunchKernelGGL((k_mesh_cells<false>), grid_for(size_t(nb) * 64, 256, 8192), dim3(256), 0, m->stream, keys.as<uint64_t>(), means.as<float>(), nb, counts.as<uint32_t>(), (uint64_t*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(k_unique_offsets, dim3(1), dim3(256), 0, m->stream, counts.as<uint32_t>(), nb, d_total.as<uint32_t>());
    uint32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_total.p, 4, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    *count = total;
    if (!codes && !corners) return CHAD_OK;
    if (capacity < total || !codes || !corners) return fail(CHAD_ERR_INVALID_ARG, "cell buffers too small");
    if (!total) return CHAD_OK;
    DevBuf d_codes, d_corners;
    HIP_TRY(d_codes.alloc(size_t(total) * 8)); HIP_TRY(d_corners.alloc(size_t(total) * 32));
    hipLaunchKernelGGL((k_mesh_cells<true>), grid_for(size_t(nb) * 64, 256, 8192), dim3(256), 0, m->stream, keys.as<uint64_t>(), means.as<float>(), nb, counts.as<uint32_t>(), d_codes.as<uint64_t>(), d_corners.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(codes, d_codes.p, size_t(total) * 8, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(corners, d_corners.p, size_t(total) * 32, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return CHAD_OK;
}

chad_status chad_map_save_mesh(chad_map* m, int32_t submap, const char* filename) {
    if (!m || !filename) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    TRY(extract_mesh(m, submap));
    std::FILE* f = std::fopen(filename, "wb");
    if (!f) return fail(CHAD_ERR_INVALID_ARG, std::string("cannot open ") + filename);
    const size_t nv = m->mesh_vertices.size() / 3, nt = m->mesh_triangles.size() / 3;
    // vertices carry their normals, as the model the reference saves does (lvr2.cpp:309-313: SimpleFinalizer with setNormalData)
    std::fprintf(f, "ply\nformat binary_little_endian 1.0\ncomment chad_tsdf mesh: sd = 0 surface of submap %d, res %g trunc %g\nelement vertex %zu\nproperty float x\nproperty float y\nproperty float z\n"
                    "property float nx\nproperty float ny\nproperty float nz\nelement face %zu\nproperty list uchar uint vertex_indices\nend_header\n", submap, double(m->res), double(m->trunc), nv, nt);
    std::vector<float> posed(nv * 6);
    for (size_t v = 0; v < nv; v++) {
        std::memcpy(&posed[6 * v], &m->mesh_vertices[3 * v], 12); std::memcpy(&posed[6 * v + 3], &m->mesh_normals[3 * v], 12);
        apply_submap_pose(m, submap, &posed[6 * v]); apply_submap_rotation(m, submap, &posed[6 * v + 3]);
    }
    bool ok = nv == 0 || std::fwrite(posed.data(), 24, nv, f) == nv;
    std::vector<unsigned char> faces(nt * 13);
    for (size_t t = 0; t < nt; t++) { faces[13 * t] = 3; std::memcpy(&faces[13 * t + 1], &m->mesh_triangles[3 * t], 12); }
    ok = ok && (nt == 0 || std::fwrite(faces.data(), 13, nt, f) == nt);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? CHAD_OK : fail(CHAD_ERR_INVALID_ARG, std::string("short write to ") + filename);
}

chad_status chad_map_last_frame(chad_map* m, uint32_t* order, uint64_t* mc, float* normals, size_t capacity, size_t* count) {
    if (!m || !count) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(m->device);
    uint32_t overflow = 0;
    TRY(read_back_header(m, &overflow));
    const FrameSlot& f = m->slot[m->last_slot];
    const size_t n = m->frames && !m->frames_empty_tail ? m->h_hdr->n_usable : 0;
    *count = n;
    const size_t k = std::min(n, capacity);
    if (!k) return CHAD_OK;
    std::vector<float> tmp(k * 4);
    if (order) {
        HIP_TRY(hipMemcpy(tmp.data(), f.sort.points[(m->h_hdr->sort_passes & 1u) ^ 1u], k * 16, hipMemcpyDeviceToHost));   // (k_frame_normals' ordered output)
        for (size_t i = 0; i < k; i++) std::memcpy(&order[i], &tmp[4 * i + 3], 4);
    }
    if (mc) {  // the codes are not kept (the frame kernels work from the sort keys): recompute them from the points
        uint64_t* d_mc = nullptr;
        HIP_TRY(chad_malloc(&d_mc, k * 8));
        hipLaunchKernelGGL(k_morton_codes_of_points, grid_for(k, 256, 1 << 22), dim3(256), 0, m->stream, reinterpret_cast<const float4*>(f.sort.points[(m->h_hdr->sort_passes & 1u) ^ 1u]),
                           uint32_t(k), m->recip, d_mc);
        HIP_TRY(hipMemcpyAsync(mc, d_mc, k * 8, hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
        (void)chad_free(d_mc);
    }
    if (normals) {
        HIP_TRY(hipMemcpy(tmp.data(), f.d_normals, k * 16, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < k; i++) std::memcpy(&normals[3 * i], &tmp[4 * i], 12);
    }
    return CHAD_OK;
}

chad_status chad_walk_rays(const float* xyz, size_t n, const float position[3], float sdf_res, float sdf_trunc, int32_t* voxels, uint32_t* counts, size_t stride) {
    if ((n && (!xyz || !voxels || !counts)) || !position || !stride) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    if (!n) return CHAD_OK;
    float* d_xyz = nullptr; int* d_vox = nullptr; uint32_t* d_cnt = nullptr;
    HIP_TRY(chad_malloc(&d_xyz, n * 12));
    HIP_TRY(chad_malloc(&d_vox, n * stride * 12));
    HIP_TRY(chad_malloc(&d_cnt, n * 4));
    HIP_TRY(hipMemcpy(d_xyz, xyz, n * 12, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_vox, 0, n * stride * 12));
    const float recip = float(1.0 / double(sdf_res));
    hipLaunchKernelGGL(k_walk_rays, dim3(unsigned((n + 255) / 256)), dim3(256), 0, 0, d_xyz, uint32_t(n), F3{position[0], position[1], position[2]}, sdf_res, recip,
                       sdf_trunc, d_vox, d_cnt, uint32_t(stride));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(voxels, d_vox, n * stride * 12, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(counts, d_cnt, n * 4, hipMemcpyDeviceToHost));
    (void)chad_free(d_xyz); (void)chad_free(d_vox); (void)chad_free(d_cnt);
    return CHAD_OK;
}

chad_status chad_sort_pairs(const uint64_t* keys, size_t n, uint32_t key_bits, uint64_t* keys_out, uint32_t* order_out) {
    if (n && !keys) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    if (n >= (1ull << 30) || key_bits > 64) return fail(CHAD_ERR_INVALID_ARG, "n or key_bits out of range");
    if (!n) return CHAD_OK;
    RadixSortBuffers b{};
    const uint32_t tiles = radix_tiles(n);
    const size_t cap = size_t(tiles) * kSortTile;
    for (int i = 0; i < 2; i++) { HIP_TRY(chad_malloc(&b.keys[i], cap * 8)); HIP_TRY(chad_malloc(&b.idx[i], cap * 4)); }
    HIP_TRY(chad_malloc(&b.hist_part, size_t(kMaxHistBlocks) * kMaxPasses * kRadix * 4));
    HIP_TRY(chad_malloc(&b.digit_base, size_t(kBaseStripes) * kMaxPasses * kRadix * 4));
    HIP_TRY(chad_malloc(&b.tile_status, radix_status_words(tiles) * 4));
    b.resident = std::getenv("CHAD_DEBUG_SORT_TICKETS") ? 0u : radix_resident_workgroups<IndexPayload>();   // nothing else runs beside this sort
    RadixSortSignals sig;
    HIP_TRY(chad_malloc(&sig.fault, 4));
    HIP_TRY(hipMemset(sig.fault, 0, 4));
    HIP_TRY(hipMemcpy(b.keys[0], keys, n * 8, hipMemcpyHostToDevice));
    const uint32_t passes = std::max(1u, (key_bits + kRadixBits - 1) / kRadixBits);
    int reps = 1;
    if (const char* r = std::getenv("CHAD_DEBUG_SORT_REPS")) reps = std::max(1, std::atoi(r));  // timing experiments
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    for (int r = 0; r < reps; r++) {
        if (r == reps - 1) HIP_TRY(hipEventRecord(e0, nullptr));
        if (r) HIP_TRY(hipMemcpyAsync(b.keys[0], keys, n * 8, hipMemcpyHostToDevice, nullptr));
        radix_sort_enqueue(b, uint32_t(n), passes, nullptr, nullptr, 0, 0, sig);
        if (r == reps - 1) HIP_TRY(hipEventRecord(e1, nullptr));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (reps > 1) { float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); std::fprintf(stderr, "chad_sort_pairs: n %zu passes %u: %.3f ms (incl. one H2D of the keys)\n", n, passes, ms); }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const int out = passes & 1;
    if (keys_out) HIP_TRY(hipMemcpy(keys_out, b.keys[out], n * 8, hipMemcpyDeviceToHost));
    if (order_out) HIP_TRY(hipMemcpy(order_out, b.idx[out], n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 2; i++) { (void)chad_free(b.keys[i]); (void)chad_free(b.idx[i]); }
    (void)chad_free(b.hist_part); (void)chad_free(b.digit_base); (void)chad_free(b.tile_status);
    uint32_t fault = 0;
    HIP_TRY(hipMemcpy(&fault, sig.fault, 4, hipMemcpyDeviceToHost));
    (void)chad_free(sig.fault);
    if (fault) return fail(CHAD_ERR_HIP, "a radix sort tile gave up waiting for its predecessors");
    return CHAD_OK;
}

chad_status chad_selftest_reciprocal(uint64_t out[4]) {
    if (!out) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DevBuf d;
    HIP_TRY(d.alloc(32));
    HIP_TRY(hipMemset(d.p, 0, 32));
    hipLaunchKernelGGL(k_selftest_reciprocal, dim3(8192), dim3(256), 0, 0, d.as<unsigned long long>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d.p, 32, hipMemcpyDeviceToHost));
    return CHAD_OK;
}

chad_status chad_selftest_wave_ops(uint64_t out[7]) {
    if (!out) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    DevBuf d;
    HIP_TRY(d.alloc(56));
    HIP_TRY(hipMemset(d.p, 0, 56));
    hipLaunchKernelGGL(k_selftest_wave_ops, dim3(512), dim3(256), 0, 0, d.as<unsigned long long>(), 64u);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d.p, 56, hipMemcpyDeviceToHost));
    return CHAD_OK;
}

chad_status chad_exchange_splitters(const uint64_t* heads, uint32_t world, uint64_t* splitters) {
    if (!heads || !splitters || world < 1 || world > uint32_t(kMaxRanks)) return fail(CHAD_ERR_INVALID_ARG, "null argument or world out of range");
    return xchg_splitters(heads, world, splitters) ? CHAD_OK : fail(CHAD_ERR_INVALID_ARG, "implausible block count in a head");
}

chad_status chad_exchange_layout(const uint64_t* matrix, const uint32_t* wide, uint32_t world, uint32_t rank, uint64_t* out) {
    if (!matrix || !wide || !out || world < 1 || world > uint32_t(kMaxRanks) || rank >= world) return fail(CHAD_ERR_INVALID_ARG, "null argument or rank / world out of range");
    ExchangePlan plan;
    if (!xchg_slice_plan(matrix, wide, world, rank, plan)) return fail(CHAD_ERR_INVALID_ARG, "implausible slice size in the matrix");
    for (uint32_t r = 0; r < world; r++) { out[5 * r] = plan.send_off[r]; out[5 * r + 1] = plan.send_bytes[r]; out[5 * r + 2] = plan.recv_off[r]; out[5 * r + 3] = plan.recv_bytes[r]; out[5 * r + 4] = plan.peer_off[r]; }
    out[5 * world] = plan.longest_send;
    return CHAD_OK;
}

chad_status chad_morton_codes(const float* xyz, size_t n, float sdf_res, uint64_t* mc) {
    if (n && (!xyz || !mc)) return fail(CHAD_ERR_INVALID_ARG, "null argument");
    if (!n) return CHAD_OK;
    float* d_xyz = nullptr; uint64_t* d_mc = nullptr;
    HIP_TRY(chad_malloc(&d_xyz, n * 12));
    HIP_TRY(chad_malloc(&d_mc, n * 8));
    HIP_TRY(hipMemcpy(d_xyz, xyz, n * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_morton_codes, dim3(unsigned((n + 255) / 256)), dim3(256), 0, 0, d_xyz, uint32_t(n), float(1.0 / double(sdf_res)), d_mc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(mc, d_mc, n * 8, hipMemcpyDeviceToHost));
    (void)chad_free(d_xyz); (void)chad_free(d_mc);
    return CHAD_OK;
}

}  // extern "C"

/* chad_hip.h -- C ABI of the MI355X (gfx950) implementation of chad::TSDFMap::insert().
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  Each entry point
 * names the reference interface it replaces (paths relative to M2-TE/chad_tsdf @ 2025-07-25).
 * The C++ class chad::TSDFMap in include/chad/tsdf.hpp is a thin pImpl over these calls, and
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Threading contract (same as the reference: single-threaded, tsdf.hpp:21-171): one host thread per
 * map at a time; distinct maps are independent.  All entry points return when the device work they
 * enqueue has been submitted; chad_map_sync / export / stats / destroy are synchronisation points.
 * No entry point ever calls exit() or abort(); failures come back as a chad_status and a message
 * retrievable with chad_last_error().
 */
#ifndef CHAD_HIP_H
#define CHAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct chad_map chad_map; /* opaque; replaces TSDFMap's private _active_octree_p/_active_submap_p/_submaps (tsdf.hpp:167-170) */

typedef int32_t chad_status;
enum {
    CHAD_OK = 0,
    CHAD_ERR_INVALID_ARG = 1, /* null handle/pointer, non-positive resolution, bad submap index */
    CHAD_ERR_NO_DEVICE = 2,   /* no HIP device, or the gfx950 code object cannot be loaded (never falls back to a CPU path) */
    CHAD_ERR_OUT_OF_MEMORY = 3, /* reference: std::runtime_error from mmap failure, virtual_array.cpp:16-20 */
    CHAD_ERR_CAPACITY = 4,    /* a single frame outran the block pool (growth happens between frames); that frame was partially dropped */
    CHAD_ERR_HIP = 5          /* any other HIP runtime failure */
};

/* Optional creation parameters (pass NULL for defaults).  Set struct_size = sizeof(chad_options). */
typedef struct chad_options {
    uint32_t struct_size;
    int32_t device;      /* HIP device ordinal; -1 = the calling thread's current device */
    uint64_t max_blocks; /* initial capacity of the active submap in 4x4x4-voxel blocks (1 044 B of HBM each: two hash slots of a 64-bit key, 64 64-bit sums and 64 counts); 0 = default (1<<21).
                            Doubles automatically between frames when half full, unless CHAD_FLAG_FIXED_CAPACITY */
    uint64_t max_points; /* initial per-frame point capacity (grows on demand); 0 = default (1<<20) */
    void* stream;        /* hipStream_t of the caller: frames wait for the work enqueued on it before the insert, and operations
                            that need the map to itself run on it; NULL = a stream owned by the map.  Frames always run on
                            streams owned by the map (three in flight) */
    uint32_t flags;      /* CHAD_FLAG_* */
    uint32_t update_mode; /* CHAD_UPDATE_* */
    uint64_t dag_pool_bytes; /* HBM reserved when the map is created for the DAG of closed submaps (NodeLevels, levels.hpp:141-199): the
                            levels' stores and look-up tables are carved out of it and grow inside it, so that a long trajectory under
                            CHAD_FLAG_DAG_ONLY never calls hipMalloc / hipFree while it runs (each a device-wide stall of 2-200 ms per GB,
                            box dependent).  The counterpart of VirtualArray's up-front reservation (virtual_array.hpp:16-19).  0 = none:
                            levels grow by hipMalloc.  When the pool is used up, growth falls back to hipMalloc (chad_pipeline_times
                            shows it).  chad_map_reset keeps the reservation.  Environment override: CHAD_DAG_POOL_MB */
} chad_options;

enum {
    CHAD_FLAG_VERBOSE = 1u, /* print the reference's five stage-timer lines per insert (morton.hpp:78,100; normals.hpp:146; octree.hpp:169; tsdf.cpp:74) */
    CHAD_FLAG_TIMING = 2u,  /* record per-stage device times with HIP events (no extra synchronisation); the frames then run one
                               after the other on one stream, so that stage times do not overlap.  chad_map_stats reports their mean
                               over the inserts since the previous stats call (at most 64) */
    CHAD_FLAG_TIME_UPDATE = 4u, /* only the TSDF-update kernel is timed (ms_oct_upd), by the begin / end time stamps of its own dispatch;
                                   frames stay pipelined (up to three in flight) */
    CHAD_FLAG_FIXED_CAPACITY = 8u, /* never grow the block pool beyond max_blocks (default: it doubles between frames when half full) */
    CHAD_FLAG_DAG_SATURATE_WEIGHTS = 16u, /* weight clusters hold min(weight, 254).  Default (off) is what the reference stores:
                                            std::max<uint8_t>(w, 255) at submap.hpp:93 makes every weight byte 0xff */
    CHAD_FLAG_TIME_KERNELS = 64u, /* every kernel of an insert is timed on its own (begin / end time stamps of its dispatch, no event
                                     records in between); frames run one after the other.  Read with chad_map_kernel_times */
    CHAD_FLAG_MESH_FROM_DAG = 128u, /* chad_map_mesh / save read a closed submap's leaves back through the DAG (8-bit signed distances), as the
                                       reference's mesh extraction does (lvr2.cpp:30-90); default: from the exact means of the dense archive */
    CHAD_FLAG_SPACE_CARVING = 256u, /* EXTENSION (upstream roadmap, README.md:59-60, not built there): after a frame's update every ray also gives
                                       the voxels it crosses before its truncation band -- the traversal of octree.hpp:100-152 on
                                       [position, p - trunc * dir], without that segment's last voxel -- one observation of +sdf_trunc, where they
                                       already hold a leaf; nothing is created.  Frames then run one after the other; not with chad_map_attach_comm */
    CHAD_FLAG_MESH_TETRAHEDRA = 512u, /* chad_map_mesh / save triangulate by marching tetrahedra (six per cell, no case table, watertight by
                                         construction: rounds 2-4's triangulation) instead of marching cubes (the default: what LVR2's boxes do) */
    CHAD_FLAG_DAG_ONLY = 32u /* a closed submap is folded into the DAG as soon as its archive is complete and its dense brick copy
                                is released: device memory then holds the active submap only (long trajectories), as the reference
                                keeps nothing but the DAG of a closed submap (tsdf.cpp:53-54).  Closed submaps are then read back with
                                chad_map_dag_leaves; chad_map_export_leaves / export_blocks of a closed submap fail with INVALID_ARG */
};
enum {
    CHAD_UPDATE_AUTO = 0,    /* chosen per frame on the device from its points-per-block ratio (default) */
    CHAD_UPDATE_SCATTER = 1, /* per-voxel global atomics: the general form, best for sparse frames */
    CHAD_UPDATE_TILED = 2    /* LDS-staged voxel tile per point block, coalesced flush: best for dense frames */
};

/* Per-frame and per-map counters; filled by chad_map_stats (synchronises). */
typedef struct chad_stats {
    uint64_t frames;          /* inserts since creation */
    uint64_t points_in;       /* points passed to the last insert */
    uint64_t points_used;     /* of those, points kept (finite, not at the sensor position) */
    uint64_t updates;         /* voxel updates of the last insert (U) */
    uint64_t blocks;          /* live 4x4x4 blocks of the active submap (reference: _node_lookup.size(), octree.hpp:187) */
    uint64_t block_capacity;
    uint64_t hash_capacity;
    uint64_t closed_submaps;  /* reference: _submaps.size(), tsdf.cpp:54 */
    uint32_t sort_passes;     /* radix passes of the last insert */
    uint32_t overflow;        /* sticky; bit 0: a capacity limit dropped updates, bit 1: a radix sort tile gave up waiting (chad_map_sync reports CHAD_ERR_HIP) */
    /* mean device milliseconds per insert since the previous stats call, by the reference's stage names;
       zero unless CHAD_FLAG_TIMING / CHAD_FLAG_VERBOSE */
    float ms_mc_calc, ms_mc_sort, ms_norm_est, ms_oct_upd, ms_total;
} chad_stats;

/* TSDFMap::TSDFMap(float sdf_res = 0.05f, float sdf_trunc = 0.1f)  -- tsdf.hpp:29, tsdf.cpp:27-31 */
chad_status chad_map_create(float sdf_res, float sdf_trunc, const chad_options* options, chad_map** out);
/* TSDFMap::~TSDFMap()  -- tsdf.cpp:32-38 */
void chad_map_destroy(chad_map* map);

/* TSDFMap::insert(const float* points_p, size_t points_count, const float* position_p)  -- tsdf.hpp:50-57 -> tsdf.cpp:39-75.
 * xyz: points_count packed x,y,z triples in HOST memory, borrowed for the duration of the call. */
chad_status chad_map_insert(chad_map* map, const float* xyz, size_t points_count, const float position[3]);
/* Same, with the points already resident in DEVICE memory of the map's device (no PCIe copy).  The call returns when the
 * work is enqueued (on internal streams: up to three frames are in flight).
 * Lifetime: the buffer must stay valid and unchanged until three further inserts WITH POINTS (one per frame slot), or a
 * synchronisation point (chad_map_sync / export / stats), have returned.  The library enforces its half: before an insert
 * reuses a frame slot it waits until the device has read the input of the frame that used the slot before (the device
 * reports that through host-mapped memory; no event, no stall when the device keeps up), which also bounds how far the host
 * runs ahead -- three frames.
 * Readiness: the points must be complete when the call is made (the frame runs on streams of the map's own, which do not
 * wait for any other stream).  If a kernel or copy still in flight on some stream produces them, use the _stream form. */
chad_status chad_map_insert_device(chad_map* map, const float* xyz_device, size_t points_count, const float position[3]);
/* Same, ordered behind `producer_stream` (a hipStream_t; NULL = the legacy default stream): the frame waits, on the device,
 * for everything enqueued on that stream so far -- the work that writes xyz_device.  One event record + one stream wait. */
chad_status chad_map_insert_device_stream(chad_map* map, const float* xyz_device, size_t points_count, const float position[3], void* producer_stream);

/* Wait for all enqueued work of this map; reports CHAD_ERR_CAPACITY if any update was dropped. */
chad_status chad_map_sync(chad_map* map);
/* Octree::clear() on the active submap and forget its poses and all closed submaps  -- octree.hpp:22-29 */
chad_status chad_map_reset(chad_map* map);

/* The submap hand-over of TSDFMap::save (tsdf.cpp:78-81): if the active submap holds at least one pose it is
 * archived as a closed submap and a fresh one is started (same effect as the 5 m roll-over of tsdf.cpp:52-57). */
chad_status chad_map_close_submap(chad_map* map);

/* The roll-over rule on its own (tsdf.cpp:50-52): would an insert at `position` finalize the active submap first, i.e. is it
 * more than 5 m from the submap's first pose?  A pure query (host arithmetic, nothing enqueued); the frame-sharded multi-GPU
 * map asks it on every rank before every frame, so that all ranks merge the submap at the same point of the stream. */
chad_status chad_map_would_roll_over(chad_map* map, const float position[3], int32_t* yes);

/* Read-back extension (the reference's iterator API is commented out, tsdf.hpp:120-161).
 * submap < 0: the active submap; 0..closed-1: a submap closed by the 5 m roll-over rule (tsdf.cpp:52-58).
 * Leaves are the voxels with weight > 0, sorted by Morton code ascending: mc as MortonCode::_value
 * (morton.hpp:21-28), sd as Leaf::_signed_distance, w as Leaf::_weight (octree.hpp:15). */
chad_status chad_map_leaf_count(chad_map* map, int32_t submap, size_t* count);
chad_status chad_map_export_leaves(chad_map* map, int32_t submap, uint64_t* mc, float* sd, uint32_t* w, size_t capacity, size_t* count);
chad_status chad_map_stats(chad_map* map, chad_stats* out);
/* Measurement: mean duration of every kernel of the insert pipeline, one entry per kernel and sort pass,
 * over the inserts since the previous call.  Needs CHAD_FLAG_TIME_KERNELS.  The reference's counterpart are its five stage timers
 * (morton.hpp:76-78,98-100; normals.hpp:144-146; octree.hpp:167-169; tsdf.cpp:72-74). */
enum { CHAD_KERNEL_BOUNDS = 0, CHAD_KERNEL_KEYS = 1, CHAD_KERNEL_SORT_PASS0 = 2 /* .. 9: radix passes 0..7 */, CHAD_KERNEL_GATHER = 10,
       CHAD_KERNEL_NORMALS = 11, CHAD_KERNEL_UPDATE = 12, CHAD_KERNEL_COUNT = 13 };
typedef struct chad_kernel_times {
    float ms[CHAD_KERNEL_COUNT];          /* mean milliseconds per launch */
    uint64_t launches[CHAD_KERNEL_COUNT]; /* launches averaged over (0: the kernel did not run) */
} chad_kernel_times;
chad_status chad_map_kernel_times(chad_map* map, chad_kernel_times* out);
/* Host time of the insert path since the last call (the counters restart): the calls of chad_map_insert* altogether; of that, the time
 * spent waiting for the device to hand a frame slot back (at most three frames are in flight: a host that runs ahead waits here -- that
 * wait is the device's pace, not host work); of that, the host-to-device copy inside chad_map_insert.  insert_s - wait_s - copy_s is
 * what the host itself needs per insert: pose bookkeeping and the launches. */
typedef struct chad_host_times { uint64_t inserts; double insert_s, wait_s, copy_s; } chad_host_times;
chad_status chad_map_host_times(chad_map* map, chad_host_times* out);
const char* chad_kernel_name(int32_t id);
/* Where a roll-over's time goes, and whether anything was allocated (measurement; the reference's counterpart is
 * the wall-clock timer around insert, tsdf.cpp:41,72-74, which includes Submap::finalize at a roll-over, tsdf.cpp:53).  Counters since
 * the previous call (they restart); set struct_size = sizeof(chad_pipeline_times) before the call -- the library fills no more than that.
 *   host side   rollover_host_s: inside the roll-over's bookkeeping (table swap, sizing, enqueuing step A; includes any wait for the
 *               roll-over before it); fold_host_s: enqueuing the stages of Submap::finalize on the device (about fifty launches per
 *               submap); fold_wait_s: BLOCKED until a fold in flight had finished (two submaps on their way, or a sync / export)
 *   device side step_a_ms: closing table -> block keys, their order, cluster words (summed over the folds completed); step_b_ms: the fold
 *               of those words into the DAG's levels.  A runs beside the B of the roll-over before it, both beside the frames
 *   allocation  hipMalloc / hipFree calls the library made on the CALLING THREAD, their bytes and host seconds; level_growths: DAG
 *               stores or tables replaced by bigger ones; dag_pool_*: chad_options.dag_pool_bytes, in use now, most in use so far
 *   exchange    sharded maps: exchanges completed, bytes this rank handed to their collectives (send side), blocks in the last union */
typedef struct chad_pipeline_times {
    uint32_t struct_size;
    uint32_t rollovers, folds_completed, level_growths;
    double rollover_host_s, fold_host_s, fold_wait_s;
    double step_a_ms, step_b_ms;
    uint64_t mallocs, frees, malloc_bytes;
    double alloc_s;
    uint64_t dag_pool_bytes, dag_pool_used, dag_pool_peak;
    uint64_t exchanges, exchange_bytes, exchange_union_blocks;
    uint32_t comm_world, comm_transport; /* ranks of the attached communicator as the TRANSPORT reports them (ncclCommCount for RCCL; 0 = none attached) / 1 RCCL, 2 callbacks */
    uint64_t comm_p2p_calls;             /* point-to-point steps of the attached communicator that went through a group of ncclSend / ncclRecv (since it was created) */
} chad_pipeline_times;
chad_status chad_map_pipeline_times(chad_map* map, chad_pipeline_times* out);

/* Multi-GPU merge support: per-block partial sums of a submap, on the DEVICE, sorted by block key
 * (key = MortonCode & ~63, the reference's lookup key, octree.hpp:33-36).  sum[b*64+v] is the sum of
 * the clamped signed distances observed by voxel v of block b, cnt[b*64+v] their count, so partials of
 * different GPUs combine by addition (the reference's update is a mean of unit-weight observations,
 * octree.hpp:161-163).  Inside the active map the sums are 64-bit integers in units of sdf_trunc / 2^30 (exact and
 * independent of the order of the updates: the same frames give the same map bit for bit); what is exported here is
 * that integer converted to f32 with one rounding, NaN where the reference's leaf is NaN. */
chad_status chad_map_block_count(chad_map* map, int32_t submap, size_t* blocks);
chad_status chad_map_export_blocks_device(chad_map* map, int32_t submap, uint64_t* keys_device, float* sum_device, uint32_t* cnt_device, size_t capacity_blocks);
/* The same blocks in no particular order (skips the sort: the multi-GPU merge orders the union of all ranks' keys anyway). */
chad_status chad_map_export_blocks_device_unordered(chad_map* map, int32_t submap, uint64_t* keys_device, float* sum_device, uint32_t* cnt_device, size_t capacity_blocks);
/* Close the active submap as if a roll-over happened, replacing its content by the given (merged) blocks. */
chad_status chad_map_close_submap_with_blocks_device(chad_map* map, const uint64_t* keys_device, const float* sum_device, const uint32_t* cnt_device, size_t blocks);

/* ---- Multi-GPU: one process per GPU, whole frames dealt round-robin, ONE exchange per submap ------------
 * The reference is a single process; what makes the path shard is its arithmetic: a leaf is the mean of unit-weight
 * observations (octree.hpp:161-163) and a frame's stages depend on that frame alone (tsdf.cpp:64-70), so per-GPU partial
 * (sum, count) bricks combine by addition -- exactly, because the map holds 64-bit integer sums: N ranks archive bit for bit
 * the submaps one GPU archives from the same stream.
 * Use: every rank creates a communicator and attaches it to its map; then EVERY rank calls the same insert / close_submap /
 * save sequence with the same positions (points are only read on the rank that owns the frame: insert k of the stream
 * belongs to rank k % world; the others may pass n = 0).  The 5 m roll-over rule (tsdf.cpp:46-61) sees every pose on every
 * rank, so all ranks reach the exchange at the same frame.  What the exchange is depends on what the map keeps of a closed submap:
 *   archives kept (default)   all-gather of block keys, sorted union, all-reduce(sum) of the i64 sums and u32 counts over the union
 *                             (RCCL over xGMI): EVERY rank archives the merged submap;
 *   CHAD_FLAG_DAG_ONLY        the sparse exchange: block keys, 64-bit voxel masks and one word per touched voxel travel point to point
 *                             (ncclSend / ncclRecv) to the owner of a Morton key range -- G - 1 splitters drawn from samples of every
 *                             rank's sorted keys --, every rank turns its range into leaf-cluster words (submap.hpp:70-99), rank 0
 *                             concatenates the ranges and folds the submap into the DAG: Submap::finalize runs ONCE per submap, and
 *                             the DAG lives on rank 0 only (chad_map_dag_* fail on the other ranks with a message).
 * Every wait of an exchange is bounded (CHAD_COMM_TIMEOUT_S, default 120 s): a peer that never arrives is an error, not a hang; a
 * failed exchange is reported once, its submap comes out empty, and the communicator is dead from then on. */
typedef struct chad_comm chad_comm;
enum { CHAD_COMM_ID_BYTES = 128 };
/* RCCL transport.  Rank 0 makes an id (ncclGetUniqueId) and hands the bytes to the other ranks over any channel the
 * application has (MPI, a file, torch.distributed ...); then every rank calls chad_comm_create_rccl (collective, with the rank's
 * GPU current).  librccl.so.1 is loaded at this point, not before.  At most 15 ranks. */
chad_status chad_comm_unique_id(uint8_t id[CHAD_COMM_ID_BYTES]);
chad_status chad_comm_create_rccl(const uint8_t id[CHAD_COMM_ID_BYTES], int32_t rank, int32_t world, chad_comm** out);
/* Transport through the caller's own collectives on HOST buffers (the library stages device data through host memory around
 * each call): for tests and rehearsals where RCCL cannot be used, e.g. several ranks sharing one GPU over gloo or MPI.
 * all_gather: recv[world * bytes] <- every rank's send[bytes] in rank order; all_reduce: buf[count] <- element-wise sum over
 * the ranks (wrapping integer addition), dtype CHAD_REDUCE_U32 or CHAD_REDUCE_I64.  Return 0 on success. */
enum { CHAD_REDUCE_U32 = 0, CHAD_REDUCE_I64 = 1 };
typedef int32_t (*chad_allgather_fn)(void* user, const void* send, void* recv, size_t bytes);
typedef int32_t (*chad_allreduce_fn)(void* user, void* buf, size_t count, int32_t dtype);
chad_status chad_comm_create_callbacks(int32_t rank, int32_t world, chad_allgather_fn all_gather, chad_allreduce_fn all_reduce, void* user, chad_comm** out);
void chad_comm_destroy(chad_comm* comm);
/* Make `map` one rank's share of a frame-sharded map (NULL detaches).  From here on chad_map_insert* integrate only the
 * frames this rank owns and record the pose of the others, and a submap is closed -- by the roll-over rule inside insert, by
 * chad_map_close_submap, by save() -- through the exchange above instead of a local archive.  The communicator must outlive
 * the attachment. */
chad_status chad_map_attach_comm(chad_map* map, chad_comm* comm);
/* The exchange on its own (collective): merge the active submaps of all ranks and archive the result on every rank.
 * comm NULL = the attached one; a one-rank communicator makes it chad_map_close_submap. */
chad_status chad_map_merge_and_close(chad_map* map, chad_comm* comm);

/* Submap::finalize (submap.hpp:10-106) and NodeLevels (levels.hpp:141-199): every closed submap is folded, in closing
 * order, into one map-wide DAG -- twenty node levels plus the leaf clusters (cluster.hpp:9-108), each storing a node once
 * and handing out addresses in order of first occurrence, the signed-distance tree before the weight tree.  The device
 * quantises each brick into its eight leaf clusters and delivers them in Morton order; the host folds that stream into
 * the levels.  Lazy: chad_map_finalize (and every chad_map_dag_* call) folds the submaps closed since the last call;
 * the roll-over inside insert() only archives the bricks, so the insert loop never waits for it.
 * The arrays are address-for-address the reference's NodeLevel::_raw_data[0, _occupied_n) (level 0..19, uint32 words:
 * child mask, then the addresses of the set bits) and LeafClusterLevel::_raw_data (level 20 = CHAD_DAG_CLUSTER_LEVEL,
 * uint64 words); word 0 of every level is reserved (levels.hpp:52,118). */
enum { CHAD_DAG_NODE_LEVELS = 20, CHAD_DAG_CLUSTER_LEVEL = 20 };
typedef struct chad_dag_stats {
    uint32_t struct_size;           /* set to sizeof(chad_dag_stats) before the call: the library fills no more than that */
    uint32_t reserved;
    uint64_t submaps;               /* closed submaps folded into the DAG so far */
    uint32_t uniques[21], dupes[21]; /* NodeLevel::_uniques_n/_dupes_n per level (levels.hpp:88); [20] = leaf clusters (levels.hpp:136) */
    float ms_device, ms_host;       /* of the last chad_map_finalize that had work: device time (key sort, cluster kernel, the fold of the levels; under CHAD_FLAG_DAG_ONLY the fold alone) /
                                     * host time (bringing the level arrays over when they are read; the whole fold under CHAD_DEBUG_DAG_HOST) */
    uint64_t device_bytes;          /* HBM held by the DAG: the levels' stores, their look-up tables, the fold's scratch, and what a level has outgrown
                                     * since the last chad_map_sync (freed there: a hipFree waits for the whole device) -- the reference keeps its
                                     * NodeLevels in host memory; here Submap::finalize runs on the device, so the DAG lives there */
} chad_dag_stats;
chad_status chad_map_finalize(chad_map* map);
/* Expected final size of the DAG's levels: entries[l] values (chad_dag_stats.uniques of an earlier run of the same kind of trajectory) and
 * words[l] (chad_map_dag_level_words), l = 0..19 the node levels, 20 the leaf clusters.  The levels' stores and look-up tables are then
 * made that size (plus a sixteenth) when the DAG is created -- the map's first fold, or the first after chad_map_reset -- and never grow
 * on the way there: no copy, no rehash, and with chad_options.dag_pool_bytes no allocation either.  The counterpart of sizing
 * VirtualArray's reservation (virtual_array.hpp:16-19) for the job at hand.  A level that outgrows its hint grows as usual -- and a
 * fold asks for room for the worst case of the submap in hand (every candidate a new value, nine words per node candidate) on top of
 * what the level holds: for small DAGs pass a hint with that room in it. */
chad_status chad_map_dag_reserve(chad_map* map, const uint64_t entries[21], const uint64_t words[21]);
/* Submap::root_addr_tsdf / root_addr_weight (submap.hpp:108-109) of closed submap `submap`: addresses into level 0 */
chad_status chad_map_dag_roots(chad_map* map, int32_t submap, uint32_t* root_tsdf, uint32_t* root_weight);
chad_status chad_map_dag_level_words(chad_map* map, uint32_t level, size_t* words);
/* out: uint32_t[capacity_words] for level 0..19, uint64_t[capacity_words] for level 20 */
chad_status chad_map_dag_level_copy(chad_map* map, uint32_t level, void* out, size_t capacity_words);
chad_status chad_map_dag_stats(chad_map* map, chad_dag_stats* out);
/* Read-back through the DAG, the way the reference's mesh extraction walks it (lvr2.cpp:30-90): the leaves of closed
 * submap `submap` ascending by Morton code, sd de-quantised from its byte (cluster.hpp:33-52).  mc / sd may be NULL. */
chad_status chad_map_dag_leaves(chad_map* map, int32_t submap, uint64_t* mc, float* sd, size_t capacity, size_t* count);
/* Stage read-back for parity tests: the device half on its own.  keys[b]: block key, ascending; tsdf[8*b+o] / weights[8*b+o]:
 * cluster words of octet o (bits 3..5 of the code) of block b, all ones where the octet holds no leaf.  Host pointers,
 * weights may be NULL; submap < 0: the active submap. */
chad_status chad_map_export_clusters(chad_map* map, int32_t submap, uint64_t* keys, uint64_t* tsdf, uint64_t* weights, size_t capacity_blocks, size_t* blocks);

/* ---- Read-back API.  The reference sketches it and leaves it commented out / TODO: a leaf iterator per
 * submap that exposes "real floating point position per leaf" and "the data written there (signed distance, weight)", never
 * Morton codes (tsdf.hpp:120-161), begin() walking the DAG to its first leaf (tsdf.cpp:104-146), operator++ / -- / + / - and
 * end() as TODO; plus "raycast to retrieve leaves along it + physics hit" (tsdf.hpp:156-158). */
typedef struct chad_leaf {
    float x, y, z;         /* position of the leaf: its voxel's corner, vec3(MortonCode::decode()) * sdf_res (octree.hpp:157, lvr2.cpp:78-79) */
    float signed_distance; /* from the DAG: de-quantised from its byte (cluster.hpp:33-52) */
    uint32_t weight;       /* the DAG's weight byte: 255 as the reference stores it (Q6, submap.hpp:93), min(w, 254) with CHAD_FLAG_DAG_SATURATE_WEIGHTS */
    uint32_t reserved;
} chad_leaf;
typedef struct chad_iter chad_iter;
/* TSDFMap::begin(root_addr) (tsdf.cpp:104-146) for closed submap `submap`: an iterator on its first leaf; leaves come in ascending
 * Morton order, the order of the reference's own DAG traversals.  Folds pending submaps into the DAG first (chad_map_finalize).
 * An iterator stays usable while further submaps are closed; chad_map_reset / destroy kill it (calls then fail / report invalid). */
chad_status chad_map_iter_begin(chad_map* map, int32_t submap, chad_iter** out);
void chad_iter_destroy(chad_iter* it);
int32_t chad_iter_valid(const chad_iter* it);                             /* 0 once moved past either end (the reference's end()) */
chad_status chad_iter_seek(chad_iter* it, int32_t to_last);               /* first leaf (0) or last leaf (1: tsdf.cpp:148-153 "walk to the last leaf node") */
chad_status chad_iter_advance(chad_iter* it, int64_t n, int64_t* moved);  /* operator++ (1), -- (-1), + / - (n); *moved = leaves actually moved (may be NULL) */
chad_status chad_iter_get(const chad_iter* it, chad_leaf* out);
chad_status chad_iter_read(chad_iter* it, chad_leaf* out, size_t capacity, size_t* count);   /* bulk: up to `capacity` leaves from the current one on, advancing */

/* Batched queries of the ACTIVE submap, on the device (the frames enqueued so far are included).  mode CHAD_QUERY_LEAF: the leaf
 * of the voxel containing each point -- Leaf::_signed_distance / _weight (octree.hpp:15), the exact mean as chad_map_export_leaves
 * reports it; CHAD_QUERY_TRILINEAR: the trilinear interpolation of the eight leaves at the corners of the cell the point lies in
 * (the cell of lvr2.cpp:93-113), weight = the smallest of the eight.  Where a leaf is missing: sd NaN, weight 0.
 * weight may be NULL.  _device: all three pointers are device memory of the map's device. */
enum { CHAD_QUERY_LEAF = 0, CHAD_QUERY_TRILINEAR = 1 };
chad_status chad_map_query_sd(chad_map* map, const float* xyz, size_t n, int32_t mode, float* sd, uint32_t* weight);
chad_status chad_map_query_sd_device(chad_map* map, const float* xyz_device, size_t n, int32_t mode, float* sd_device, uint32_t* weight_device);
/* Raycast (tsdf.hpp:157): ray i = origins[i] + t * directions[i] (unit directions), marched in steps of sdf_res / 2 through the
 * trilinear field of the active submap; t_hit[i] = distance of the first crossing from positive to non-positive signed distance in
 * [0, t_max] (linear interpolation between the two samples), NaN if there is none.  Host pointers. */
chad_status chad_map_raycast(chad_map* map, const float* origins, const float* directions, size_t n, float t_max, float* t_hit);

/* ---- Mesh: what TSDFMap::save produces through LVR2 in the reference (tsdf.cpp:85 -> lvr2.cpp:275-321).
 * The grid is the reference's (lvr2.cpp:15-133: a query point per leaf at its voxel corner, cells = cubes of eight query points,
 * cells with a missing corner culled); the sd = 0 surface inside each cell is triangulated on the device by MARCHING CUBES as LVR2's
 * boxes do it (lvr2.cpp:246-251: BoxT::getSurface per cell) -- case index bit i = corner i's distance is positive, the published
 * 256-case table, corners placed by linear interpolation along their cell edge -- or, with CHAD_FLAG_MESH_TETRAHEDRA, by marching
 * tetrahedra.  LVR2 is not available here: this is agreement with the published algorithm (held against the oracle's restatement of
 * it), not with LVR2's output, and BilinearFastBox::optimizePlanarFaces (lvr2.cpp:262) is not modelled -- the row stays parity-unpinned.
 * submap >= 0: a closed submap (the reference meshes _submaps.front()); < 0: the active one.  Vertices: 3 floats each, welded
 * (every vertex once); triangles: 3 vertex indices each, oriented towards increasing signed distance (the sensor's side).
 * Call with NULL buffers for the sizes. */
chad_status chad_map_mesh(chad_map* map, int32_t submap, float* vertices, size_t vertex_capacity, uint32_t* triangles, size_t triangle_capacity,
                          size_t* vertex_count, size_t* triangle_count);
/* ... and a unit normal per vertex of that mesh (3 floats each, the order of chad_map_mesh's vertices): the normalised sum of the unit
 * normals of the faces around the vertex -- what the reference stores with the model it saves (lvr2.cpp:296-297, 309-313). */
chad_status chad_map_mesh_normals(chad_map* map, int32_t submap, float* normals, size_t vertex_capacity, size_t* vertex_count);
chad_status chad_map_save_mesh(chad_map* map, int32_t submap, const char* filename);   /* binary little-endian PLY: x y z nx ny nz per vertex, triangles */
/* Stage read-back for parity: the grid itself, i.e. what ChadGrid's constructor (lvr2.cpp:15-133) leaves in m_cells / m_queryPoints before
 * LVR2 triangulates it -- the COMPLETE cells (all eight corners hold a leaf, lvr2.cpp:115-129), ascending by the Morton code of their
 * lowest corner (MortonCode(cell_chunk)._value, lvr2.cpp:105); corners[8 * c + i] = signed distance of the query point at vertex i
 * of cell c, vertex order of lvr2.cpp:92-102 (0 (0,0,0), 1 (1,0,0), 2 (1,1,0), 3 (0,1,0), 4..7 the same at z + 1).  Values as the mesh
 * takes them: the exact means, or -- CHAD_FLAG_MESH_FROM_DAG / CHAD_FLAG_DAG_ONLY -- the DAG's de-quantised bytes (cluster.hpp:33-52),
 * which is what the reference meshes.  Host pointers; call with NULL buffers for the count. */
chad_status chad_map_mesh_cells(chad_map* map, int32_t submap, uint64_t* codes, float* corners, size_t capacity, size_t* count);

/* ---- Loop-closure support (upstream's roadmap lists "Loop closure" as open, README.md:59, and keeps what it would
 * need: the poses of a submap's frames, Submap::positions, submap.hpp:110).  EXTENSION, nothing upstream to compare with.
 * chad_map_submap_positions: those poses, 3 floats per frame, in insertion order (submap < 0: the active submap) -- what a pose-graph
 * back-end matches.  chad_map_set_submap_pose: its answer, a rigid correction x' = R x + t (R row-major) of a CLOSED submap; the
 * iterator's leaf positions and the mesh's vertices of that submap come out corrected (the stored voxels do not move). */
chad_status chad_map_submap_positions(chad_map* map, int32_t submap, float* xyz, size_t capacity, size_t* count);
chad_status chad_map_set_submap_pose(chad_map* map, int32_t submap, const float rotation[9], const float translation[3]);

/* Stage read-back for parity tests: intermediates of the LAST insert, copied to host.
 * order[i]: input index of the i-th point in processing order; mc[i]: its Morton code; normals: 3 floats per point.
 * Any pointer may be NULL. */
chad_status chad_map_last_frame(chad_map* map, uint32_t* order, uint64_t* mc, float* normals, size_t capacity, size_t* count);
/* Stage entry for parity tests: the voxel traversal of octree.hpp:86-152 for n rays (host in/out).
 * voxels: n*stride*3 int32, counts: n. */
chad_status chad_walk_rays(const float* xyz, size_t n, const float position[3], float sdf_res, float sdf_trunc, int32_t* voxels, uint32_t* counts, size_t stride);
/* Stage entry for parity tests: MortonCode::encode of calc_morton_vector (morton.hpp:59-80) for n host points. */
chad_status chad_morton_codes(const float* xyz, size_t n, float sdf_res, uint64_t* mc);

/* Stage entry for parity tests: the stable ascending radix sort that replaces std::sort in sort_morton_vector
 * (morton.hpp:81-102).  Sorts n host keys by their low key_bits bits; keys_out[i] is the i-th smallest key and
 * order_out[i] its position in the input (equal keys keep input order).  Either output may be NULL. */
chad_status chad_sort_pairs(const uint64_t* keys, size_t n, uint32_t key_bits, uint64_t* keys_out, uint32_t* order_out);

/* Self-test: the reciprocal the ray set-up uses in place of hipcc's IEEE division sequence (v_rcp_f32 + one Newton step in FMAs for
 * operands in [2^-126, 2^126), the IEEE sequence elsewhere: chad_tsdf_amd/csrc/device_math.hpp) against 1.0f / x for ALL 2^32 f32 bit
 * patterns, on the device.  out[0] = inputs inside the fast range, out[1] = of those, results that differ (must be 0), out[2] = inputs of
 * the whole f32 range for which the function the kernels call differs from the quotient (must be 0), out[3] = inputs OUTSIDE the range
 * for which the fast sequence alone would be wrong (why the range test exists).  Bit-exact voxel lists (octree.hpp:92-118) rest on this. */
chad_status chad_selftest_reciprocal(uint64_t out[4]);

/* Self-test: the wave-wide prefix sums, suffix minima and reductions of the frame kernels and the sort, which run on the DPP network of a
 * 64-lane wave (row shifts, row_bcast:15, row_bcast:31: chad_tsdf_amd/csrc/radix_sort.hpp wave_inclusive_sum, frame_kernels.hpp wave_reduce),
 * against plain loops over the same values, on the device: 2 048 waves x 64 rounds.  out[0..5] = lanes whose inclusive sum / suffix minimum /
 * minimum / maximum / sum / 64-bit minimum differed (all must be 0), out[6] = lanes checked.  The order of a frame's points inside a block
 * (normals.hpp:81-148) and the stability of the sort (morton.hpp:81-102) rest on these. */
chad_status chad_selftest_wave_ops(uint64_t out[7]);

/* Stage entries for the CPU tests of the sparse multi-GPU exchange (no device needed): the host arithmetic every rank runs on the words
 * it has all-gathered (chad_tsdf_amd/csrc/exchange_plan.hpp).  chad_exchange_splitters: heads[world][65] -- per rank its block count
 * (low 32 bits of word 0; bit 32: two payload words per voxel) and 64 sample keys of its sorted list -- -> the world - 1 splitter keys
 * that cut the Morton key space into the ranks' ranges.  chad_exchange_layout: matrix[a][b] = {blocks, words} rank a holds for rank b's
 * range, wide[a] its payload width flag -> for rank `rank`, out[5 r .. 5 r + 4] = {send offset, send bytes, receive offset, receive
 * bytes, offset of this rank's slice in rank r's send buffer}, out[5 world] = the longest send buffer of the job. */
chad_status chad_exchange_splitters(const uint64_t* heads, uint32_t world, uint64_t* splitters);
chad_status chad_exchange_layout(const uint64_t* matrix, const uint32_t* wide, uint32_t world, uint32_t rank, uint64_t* out);

/* Thread-local message of the last failure on this thread ("" if none). */
const char* chad_last_error(void);
/* Library/build identification, e.g. "chad_hip 0.1 gfx950". */
const char* chad_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CHAD_HIP_H */

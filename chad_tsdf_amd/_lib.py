"""ctypes binding of chad_tsdf_amd/lib/libchad_hip.so (the C ABI of include/chad_hip.h).

There is no fallback: if the library is missing or cannot be loaded, importing the binding raises.
`import torch` happens first on purpose -- PyTorch ships its own libamdhip64.so.7, and loading it first makes
the dynamic linker resolve our library's HIP dependency to the same runtime, so device pointers, streams and
torch tensors are interchangeable in one process.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libchad_hip.so")

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_OUT_OF_MEMORY, ERR_CAPACITY, ERR_HIP = range(6)
FLAG_VERBOSE, FLAG_TIMING, FLAG_TIME_UPDATE, FLAG_FIXED_CAPACITY, FLAG_DAG_SATURATE_WEIGHTS, FLAG_DAG_ONLY, FLAG_TIME_KERNELS, FLAG_MESH_FROM_DAG, FLAG_SPACE_CARVING, FLAG_MESH_TETRAHEDRA = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
QUERY_LEAF, QUERY_TRILINEAR = 0, 1
KERNEL_COUNT = 13
DAG_NODE_LEVELS, DAG_CLUSTER_LEVEL = 20, 20
UPDATE_AUTO, UPDATE_SCATTER, UPDATE_TILED = 0, 1, 2
COMM_ID_BYTES, REDUCE_U32, REDUCE_I64 = 128, 0, 1
ALLGATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)   # chad_allgather_fn
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32)    # chad_allreduce_fn
MAX_SLOTS = 8  # chad_map::kMaxSlots: the most frames a map ever keeps in flight (three by default)


class ChadError(RuntimeError):
    def __init__(self, status: int, where: str, message: str):
        super().__init__(f"{where}: {message} (chad_status {status})")
        self.status = status


class Options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("max_blocks", C.c_uint64),
        ("max_points", C.c_uint64),
        ("stream", C.c_void_p),
        ("flags", C.c_uint32),
        ("update_mode", C.c_uint32),
        ("dag_pool_bytes", C.c_uint64),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("frames", C.c_uint64),
        ("points_in", C.c_uint64),
        ("points_used", C.c_uint64),
        ("updates", C.c_uint64),
        ("blocks", C.c_uint64),
        ("block_capacity", C.c_uint64),
        ("hash_capacity", C.c_uint64),
        ("closed_submaps", C.c_uint64),
        ("sort_passes", C.c_uint32),
        ("overflow", C.c_uint32),
        ("ms_mc_calc", C.c_float),
        ("ms_mc_sort", C.c_float),
        ("ms_norm_est", C.c_float),
        ("ms_oct_upd", C.c_float),
        ("ms_total", C.c_float),
    ]


class Leaf(C.Structure):   # chad_leaf
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("signed_distance", C.c_float), ("weight", C.c_uint32), ("reserved", C.c_uint32)]


class KernelTimes(C.Structure):
    _fields_ = [("ms", C.c_float * KERNEL_COUNT), ("launches", C.c_uint64 * KERNEL_COUNT)]


class DagStats(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("reserved", C.c_uint32),
        ("submaps", C.c_uint64),
        ("uniques", C.c_uint32 * 21),
        ("dupes", C.c_uint32 * 21),
        ("ms_device", C.c_float),
        ("ms_host", C.c_float),
        ("device_bytes", C.c_uint64),
    ]


class PipelineTimes(C.Structure):   # chad_pipeline_times
    _fields_ = [
        ("struct_size", C.c_uint32), ("rollovers", C.c_uint32), ("folds_completed", C.c_uint32), ("level_growths", C.c_uint32),
        ("rollover_host_s", C.c_double), ("fold_host_s", C.c_double), ("fold_wait_s", C.c_double),
        ("step_a_ms", C.c_double), ("step_b_ms", C.c_double),
        ("mallocs", C.c_uint64), ("frees", C.c_uint64), ("malloc_bytes", C.c_uint64),
        ("alloc_s", C.c_double),
        ("dag_pool_bytes", C.c_uint64), ("dag_pool_used", C.c_uint64), ("dag_pool_peak", C.c_uint64),
        ("exchanges", C.c_uint64), ("exchange_bytes", C.c_uint64), ("exchange_union_blocks", C.c_uint64),
        ("comm_world", C.c_uint32), ("comm_transport", C.c_uint32),
        ("comm_p2p_calls", C.c_uint64),
    ]


class HostTimes(C.Structure):
    _fields_ = [("inserts", C.c_uint64), ("insert_s", C.c_double), ("wait_s", C.c_double), ("copy_s", C.c_double)]


# every symbol include/chad_hip.h declares: name -> (restype, argtypes)
_vp, _sz, _i32, _f32 = C.c_void_p, C.c_size_t, C.c_int32, C.c_float
SYMBOLS = {
    "chad_map_create": (_i32, [_f32, _f32, C.POINTER(Options), C.POINTER(_vp)]),
    "chad_map_destroy": (None, [_vp]),
    "chad_map_insert": (_i32, [_vp, _vp, _sz, _vp]),
    "chad_map_insert_device": (_i32, [_vp, _vp, _sz, _vp]),
    "chad_map_insert_device_stream": (_i32, [_vp, _vp, _sz, _vp, _vp]),
    "chad_map_sync": (_i32, [_vp]),
    "chad_map_reset": (_i32, [_vp]),
    "chad_map_would_roll_over": (_i32, [_vp, _vp, C.POINTER(_i32)]),
    "chad_map_close_submap": (_i32, [_vp]),
    "chad_map_leaf_count": (_i32, [_vp, _i32, C.POINTER(_sz)]),
    "chad_map_export_leaves": (_i32, [_vp, _i32, _vp, _vp, _vp, _sz, C.POINTER(_sz)]),
    "chad_map_stats": (_i32, [_vp, C.POINTER(Stats)]),
    "chad_map_kernel_times": (_i32, [_vp, C.POINTER(KernelTimes)]),
    "chad_kernel_name": (C.c_char_p, [_i32]),
    "chad_map_block_count": (_i32, [_vp, _i32, C.POINTER(_sz)]),
    "chad_map_export_blocks_device": (_i32, [_vp, _i32, _vp, _vp, _vp, _sz]),
    "chad_map_export_blocks_device_unordered": (_i32, [_vp, _i32, _vp, _vp, _vp, _sz]),
    "chad_map_close_submap_with_blocks_device": (_i32, [_vp, _vp, _vp, _vp, _sz]),
    "chad_comm_unique_id": (_i32, [_vp]),
    "chad_comm_create_rccl": (_i32, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "chad_comm_create_callbacks": (_i32, [_i32, _i32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "chad_comm_destroy": (None, [_vp]),
    "chad_map_attach_comm": (_i32, [_vp, _vp]),
    "chad_map_merge_and_close": (_i32, [_vp, _vp]),
    "chad_map_finalize": (_i32, [_vp]),
    "chad_map_dag_reserve": (_i32, [_vp, _vp, _vp]),
    "chad_map_dag_roots": (_i32, [_vp, _i32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "chad_map_dag_level_words": (_i32, [_vp, C.c_uint32, C.POINTER(_sz)]),
    "chad_map_dag_level_copy": (_i32, [_vp, C.c_uint32, _vp, _sz]),
    "chad_map_dag_stats": (_i32, [_vp, C.POINTER(DagStats)]),
    "chad_map_host_times": (_i32, [_vp, C.POINTER(HostTimes)]),
    "chad_map_pipeline_times": (_i32, [_vp, C.POINTER(PipelineTimes)]),
    "chad_map_dag_leaves": (_i32, [_vp, _i32, _vp, _vp, _sz, C.POINTER(_sz)]),
    "chad_map_export_clusters": (_i32, [_vp, _i32, _vp, _vp, _vp, _sz, C.POINTER(_sz)]),
    "chad_map_iter_begin": (_i32, [_vp, _i32, C.POINTER(_vp)]),
    "chad_iter_destroy": (None, [_vp]),
    "chad_iter_valid": (_i32, [_vp]),
    "chad_iter_seek": (_i32, [_vp, _i32]),
    "chad_iter_advance": (_i32, [_vp, C.c_int64, C.POINTER(C.c_int64)]),
    "chad_iter_get": (_i32, [_vp, C.POINTER(Leaf)]),
    "chad_iter_read": (_i32, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "chad_map_query_sd": (_i32, [_vp, _vp, _sz, _i32, _vp, _vp]),
    "chad_map_query_sd_device": (_i32, [_vp, _vp, _sz, _i32, _vp, _vp]),
    "chad_map_raycast": (_i32, [_vp, _vp, _vp, _sz, _f32, _vp]),
    "chad_map_mesh": (_i32, [_vp, _i32, _vp, _sz, _vp, _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    "chad_map_mesh_normals": (_i32, [_vp, _i32, _vp, _sz, C.POINTER(_sz)]),
    "chad_map_mesh_cells": (_i32, [_vp, _i32, _vp, _vp, _sz, C.POINTER(_sz)]),
    "chad_map_save_mesh": (_i32, [_vp, _i32, C.c_char_p]),
    "chad_map_submap_positions": (_i32, [_vp, _i32, _vp, _sz, C.POINTER(_sz)]),
    "chad_map_set_submap_pose": (_i32, [_vp, _i32, _vp, _vp]),
    "chad_map_last_frame": (_i32, [_vp, _vp, _vp, _vp, _sz, C.POINTER(_sz)]),
    "chad_walk_rays": (_i32, [_vp, _sz, _vp, _f32, _f32, _vp, _vp, _sz]),
    "chad_morton_codes": (_i32, [_vp, _sz, _f32, _vp]),
    "chad_sort_pairs": (_i32, [_vp, _sz, C.c_uint32, _vp, _vp]),
    "chad_selftest_reciprocal": (_i32, [_vp]),
    "chad_selftest_wave_ops": (_i32, [_vp]),
    "chad_exchange_splitters": (_i32, [_vp, C.c_uint32, _vp]),
    "chad_exchange_layout": (_i32, [_vp, _vp, C.c_uint32, C.c_uint32, _vp]),
    "chad_last_error": (C.c_char_p, []),
    "chad_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libchad_hip.so (once).  Raises ImportError with the build hint if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()); there is no CPU fallback")
        import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first, see module docstring)

        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status: int, where: str):
    if status != OK:
        raise ChadError(status, where, load().chad_last_error().decode(errors="replace"))

"""ctypes binding of oracle/_build/libchad_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package
(chad_tsdf_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SANITIZE = os.environ.get("CHAD_SANITIZE", "") not in ("", "0")   # `make asan`: the ASan + UBSan build (the interpreter runs with libasan preloaded)
_SO = os.path.join(_HERE, "_build", "libchad_oracle_asan.so" if SANITIZE else "libchad_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "chad_oracle.cpp")
    if force or not os.path.exists(_SO) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if SANITIZE else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, f32, sz = C.c_void_p, C.c_float, C.c_size_t
        L.oracle_map_create.restype = vp
        L.oracle_map_create.argtypes = [f32, f32]
        L.oracle_map_destroy.argtypes = [vp]
        L.oracle_map_insert.argtypes = [vp, vp, sz, vp]
        L.oracle_map_reset.argtypes = [vp]
        L.oracle_map_closed_count.restype = sz
        L.oracle_map_closed_count.argtypes = [vp]
        L.oracle_map_leaf_count.restype = sz
        L.oracle_map_leaf_count.argtypes = [vp, C.c_int]
        L.oracle_map_export_leaves.restype = sz
        L.oracle_map_export_leaves.argtypes = [vp, C.c_int, vp, vp, vp, sz]
        L.oracle_map_stats.argtypes = [vp, vp]
        L.oracle_map_export_exact_means.restype = sz
        L.oracle_map_export_exact_means.argtypes = [vp, vp, sz]
        L.oracle_morton_codes.argtypes = [vp, sz, f32, vp]
        L.oracle_morton_encode_ivec.argtypes = [vp, sz, vp]
        L.oracle_morton_decode_ivec.argtypes = [vp, sz, vp]
        L.oracle_morton_encode_portable.restype = C.c_uint64
        L.oracle_morton_encode_portable.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.oracle_frame_normals.restype = sz
        L.oracle_frame_normals.argtypes = [vp, sz, vp, f32, vp, vp, vp, vp]
        L.oracle_walk_ray.restype = sz
        L.oracle_walk_ray.argtypes = [vp, vp, f32, f32, vp, sz]
        L.oracle_walk_rays.argtypes = [vp, sz, vp, f32, f32, vp, vp, sz]
        L.oracle_map_set_saturate_weights.argtypes = [vp, C.c_int]
        L.oracle_map_set_carving.argtypes = [vp, C.c_int]
        L.oracle_map_close_submap.argtypes = [vp]
        L.oracle_dag_level_words.restype = sz
        L.oracle_dag_level_words.argtypes = [vp, C.c_uint32]
        L.oracle_dag_level_copy.argtypes = [vp, C.c_uint32, vp]
        L.oracle_dag_level_counts.argtypes = [vp, C.c_uint32, vp]
        L.oracle_dag_roots.restype = C.c_int
        L.oracle_dag_roots.argtypes = [vp, C.c_uint32, vp]
        L.oracle_map_finalize_ms.restype = C.c_double
        L.oracle_map_finalize_ms.argtypes = [vp]
        L.oracle_map_close_submap_from_leaves.argtypes = [vp, vp, vp, vp, sz]
        L.oracle_chad_grid.restype = sz
        L.oracle_chad_grid.argtypes = [vp, C.c_uint32, vp, vp, sz, vp]
        L.oracle_marching_cubes.restype = sz
        L.oracle_marching_cubes.argtypes = [vp, vp, sz, C.c_float, vp, sz]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


STAT_NAMES = ("ms_calc", "ms_sort", "ms_norm", "ms_upd", "ms_total", "updates", "leaves", "blocks", "nodes", "frames")


class OracleMap:
    """CPU restatement of chad::TSDFMap (tsdf.hpp:21-171) with a leaf export for parity checks."""

    def __init__(self, res: float = 0.05, trunc: float = 0.1):
        self._h = lib().oracle_map_create(res, trunc)
        self.res, self.trunc = res, trunc

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oracle_map_destroy(self._h)
            self._h = None

    def insert(self, points, position):
        pts = _f32(points).reshape(-1, 3)
        pos = _f32(position).reshape(3)
        lib().oracle_map_insert(self._h, _p(pts), pts.shape[0], _p(pos))

    def reset(self):
        lib().oracle_map_reset(self._h)

    @property
    def closed_submaps(self) -> int:
        return lib().oracle_map_closed_count(self._h)

    def export_leaves(self, submap: int = -1):
        n = lib().oracle_map_leaf_count(self._h, submap)
        mc = np.empty(n, dtype=np.uint64)
        sd = np.empty(n, dtype=np.float32)
        w = np.empty(n, dtype=np.uint32)
        got = lib().oracle_map_export_leaves(self._h, submap, _p(mc), _p(sd), _p(w), n)
        assert got == n
        return mc, sd, w

    # ---- Submap::finalize / NodeLevels (submap.hpp:10-106, levels.hpp) ----
    def set_saturate_weights(self, on: bool):
        lib().oracle_map_set_saturate_weights(self._h, int(on))

    def set_carving(self, on: bool):
        """The space-carving extension (not in the reference): see carve_frame in chad_oracle.cpp."""
        lib().oracle_map_set_carving(self._h, int(on))

    def close_submap(self):
        """TSDFMap::save's hand-over (tsdf.cpp:78-81): finalize + archive the active submap if it holds a pose."""
        lib().oracle_map_close_submap(self._h)

    def close_submap_from_leaves(self, mc, sd, w):
        mc = np.ascontiguousarray(mc, dtype=np.uint64)
        sd = _f32(sd)
        w = np.ascontiguousarray(w, dtype=np.uint32)
        lib().oracle_map_close_submap_from_leaves(self._h, _p(mc), _p(sd), _p(w), mc.shape[0])

    def dag_level(self, level: int) -> np.ndarray:
        """Raw words of NodeLevels::_nodes[level] (u32, level 0..19) or of the leaf clusters (u64, level 20)."""
        n = lib().oracle_dag_level_words(self._h, level)
        out = np.empty(n, dtype=np.uint64 if level == 20 else np.uint32)
        lib().oracle_dag_level_copy(self._h, level, _p(out))
        return out

    def dag_level_counts(self, level: int):
        out = np.zeros(2, dtype=np.uint32)
        lib().oracle_dag_level_counts(self._h, level, _p(out))
        return int(out[0]), int(out[1])

    def dag_roots(self, submap: int):
        out = np.zeros(2, dtype=np.uint32)
        if lib().oracle_dag_roots(self._h, submap, _p(out)):
            raise IndexError(submap)
        return int(out[0]), int(out[1])

    def chad_grid(self, submap: int):
        """ChadGrid of a closed submap (lvr2.cpp:15-133): (codes u64 [C] ascending -- Morton code of each complete cell's lowest corner --,
        corners f32 [C, 8] -- the query points' signed distances in the cell's vertex order, lvr2.cpp:92-102 --, number of query points)."""
        qp = np.zeros(1, dtype=np.uint64)
        n = lib().oracle_chad_grid(self._h, submap, None, None, 0, _p(qp))
        codes = np.empty(n, dtype=np.uint64)
        corners = np.empty((n, 8), dtype=np.float32)
        lib().oracle_chad_grid(self._h, submap, _p(codes), _p(corners), n, _p(qp))
        return codes, corners, int(qp[0])

    @staticmethod
    def marching_cubes(codes: np.ndarray, corners: np.ndarray, res: float) -> np.ndarray:
        """Triangle soup f32 [T, 3, 3] of marching cubes over complete cells (codes u64 [C], corners f32 [C, 8]): the published algorithm LVR2's
        FastBox implements (case bit i = corner i's distance positive, the 256-case table, linear interpolation along cell edges)."""
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        corners = np.ascontiguousarray(corners, dtype=np.float32)
        n = lib().oracle_marching_cubes(_p(codes), _p(corners), len(codes), res, None, 0)
        out = np.empty((n, 3, 3), dtype=np.float32)
        assert lib().oracle_marching_cubes(_p(codes), _p(corners), len(codes), res, _p(out), n) == n
        return out

    @property
    def finalize_ms(self) -> float:
        return lib().oracle_map_finalize_ms(self._h)

    def exact_means(self) -> np.ndarray:
        """f64 mean of the clamped distances of every leaf of the active octree (order of export_leaves(-1)): what the
        reference's f32 running mean (octree.hpp:161-163) approximates."""
        n = lib().oracle_map_leaf_count(self._h, -1)
        out = np.empty(n, dtype=np.float64)
        assert lib().oracle_map_export_exact_means(self._h, _p(out), n) == n
        return out

    def stats(self) -> dict:
        out = np.zeros(10, dtype=np.float64)
        lib().oracle_map_stats(self._h, _p(out))
        return dict(zip(STAT_NAMES, out.tolist()))


def morton_codes(points, res: float) -> np.ndarray:
    pts = _f32(points).reshape(-1, 3)
    out = np.empty(pts.shape[0], dtype=np.uint64)
    lib().oracle_morton_codes(_p(pts), pts.shape[0], res, _p(out))
    return out


def morton_encode_ivec(v) -> np.ndarray:
    v = np.ascontiguousarray(v, dtype=np.int32).reshape(-1, 3)
    out = np.empty(v.shape[0], dtype=np.uint64)
    lib().oracle_morton_encode_ivec(_p(v), v.shape[0], _p(out))
    return out


def morton_decode_ivec(mc) -> np.ndarray:
    mc = np.ascontiguousarray(mc, dtype=np.uint64)
    out = np.empty((mc.shape[0], 3), dtype=np.int32)
    lib().oracle_morton_decode_ivec(_p(mc), mc.shape[0], _p(out))
    return out


def frame_normals(points, position, res: float):
    """Stages a4-a6: (order, sorted codes, normals per sorted index, group-start flags)."""
    pts = _f32(points).reshape(-1, 3)
    pos = _f32(position).reshape(3)
    n = pts.shape[0]
    order = np.empty(n, dtype=np.uint32)
    mc = np.empty(n, dtype=np.uint64)
    nrm = np.empty((n, 3), dtype=np.float32)
    gs = np.zeros(n, dtype=np.uint8)
    k = lib().oracle_frame_normals(_p(pts), n, _p(pos), res, _p(order), _p(mc), _p(nrm), _p(gs))
    return order[:k], mc[:k], nrm[:k], gs[:k]


def walk_rays(points, position, res: float, trunc: float, stride: int = 32):
    """Stage a8: per-ray voxel lists, int32 [n, stride, 3] and counts [n]."""
    pts = _f32(points).reshape(-1, 3)
    pos = _f32(position).reshape(3)
    n = pts.shape[0]
    vox = np.zeros((n, stride, 3), dtype=np.int32)
    cnt = np.zeros(n, dtype=np.uint32)
    lib().oracle_walk_rays(_p(pts), n, _p(pos), res, trunc, _p(vox), _p(cnt), stride)
    return vox, cnt

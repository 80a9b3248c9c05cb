"""Host-side mirror of chad::TSDFMap (reference include/chad/tsdf.hpp:21-171) over the C ABI.

Same names and argument meaning as the reference class: TSDFMap(sdf_res=0.05, sdf_trunc=0.1),
map.insert(points, position), map.save(filename).  Points are accepted as anything numpy can view as
float32 [N,3] (host memory, copied over PCIe by the library) or as a torch CUDA tensor (device-resident,
no copy).  All compute happens in the HIP library; nothing here touches the oracle.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _np_f32(a, shape_last=3):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(-1, shape_last) if a.size else a.reshape(0, shape_last)


class TSDFMap:
    def __init__(self, sdf_res: float = 0.05, sdf_trunc: float = 0.1, *, device: int = -1, max_blocks: int = 0,
                 max_points: int = 0, stream: int = 0, verbose: bool = False, timing: bool = False, time_update: bool = False, fixed_capacity: bool = False, update_mode: int = 0,
                 dag_saturate_weights: bool = False, dag_only: bool = False, time_kernels: bool = False, mesh_from_dag: bool = False, space_carving: bool = False,
                 dag_pool_bytes: int = 0, mesh_tetrahedra: bool = False):
        self._lib = _lib.load()
        self._sdf_res = float(sdf_res)
        self._sdf_trunc = float(sdf_trunc)
        opt = _lib.Options()
        opt.struct_size = C.sizeof(_lib.Options)
        opt.device = device
        opt.max_blocks = max_blocks
        opt.max_points = max_points
        opt.stream = stream or None
        opt.flags = (_lib.FLAG_VERBOSE if verbose else 0) | (_lib.FLAG_TIMING if timing else 0) | (_lib.FLAG_TIME_UPDATE if time_update else 0) | (_lib.FLAG_FIXED_CAPACITY if fixed_capacity else 0) | (_lib.FLAG_DAG_SATURATE_WEIGHTS if dag_saturate_weights else 0) | (_lib.FLAG_DAG_ONLY if dag_only else 0) | (_lib.FLAG_TIME_KERNELS if time_kernels else 0) | (_lib.FLAG_MESH_FROM_DAG if mesh_from_dag else 0) | (_lib.FLAG_SPACE_CARVING if space_carving else 0) | (_lib.FLAG_MESH_TETRAHEDRA if mesh_tetrahedra else 0)
        opt.update_mode = update_mode
        opt.dag_pool_bytes = dag_pool_bytes
        handle = C.c_void_p()
        _lib.check(self._lib.chad_map_create(sdf_res, sdf_trunc, C.byref(opt), C.byref(handle)), "TSDFMap")
        self._h = handle
        self._yes = C.c_int32()
        self._yes_ref = C.byref(self._yes)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.chad_map_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- reference API ------------------------------------------------------------------------------
    def insert(self, points, position, ready: bool = False):
        """TSDFMap::insert(points, position) -- tsdf.hpp:48-64.

        A torch CUDA tensor is taken zero-copy.  The frame is ordered behind torch's current stream (whatever produces the
        tensor there -- a cast, a kernel, a copy -- comes first) unless ready=True: the caller then vouches that the data is
        complete (e.g. after torch.cuda.synchronize()), which saves an event record and a stream wait per insert."""
        pos = _np_f32(position).reshape(3)
        if hasattr(points, "is_cuda") and points.is_cuda:  # torch tensor on the GPU: zero-copy
            import torch

            t = points.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(torch.float32).contiguous()
            n = t.numel() // 3
            if ready:
                _lib.check(self._lib.chad_map_insert_device(self._h, t.data_ptr(), n, pos.ctypes.data), "TSDFMap.insert")
            else:
                _lib.check(self._lib.chad_map_insert_device_stream(self._h, t.data_ptr(), n, pos.ctypes.data, torch.cuda.current_stream(t.device).cuda_stream),
                           "TSDFMap.insert")
            # The buffer must outlive the kernels that read it: until one further insert per frame slot has returned
            # (include/chad_hip.h; the library waits for the device there).  Holding the tensors also keeps torch's caching
            # allocator from handing the block to somebody else's kernel on another stream meanwhile.
            if n:
                self._keepalive = (getattr(self, "_keepalive", ()) + (t,))[-_lib.MAX_SLOTS:]
        else:
            pts = _np_f32(points)
            _lib.check(self._lib.chad_map_insert(self._h, pts.ctypes.data, pts.shape[0], pos.ctypes.data), "TSDFMap.insert")

    def attach_comm(self, comm):
        """Make this map one rank's share of a frame-sharded map (chad_map_attach_comm); comm: chad_tsdf_amd.sharded.Comm or None."""
        _lib.check(self._lib.chad_map_attach_comm(self._h, comm.handle if comm is not None else None), "TSDFMap.attach_comm")
        self._comm = comm   # must outlive the attachment

    def merge_and_close(self, comm=None):
        """The exchange step (collective over `comm`, default: the attached communicator): merge the ranks' active submaps,
        archive the result on every rank."""
        _lib.check(self._lib.chad_map_merge_and_close(self._h, comm.handle if comm is not None else None), "TSDFMap.merge_and_close")

    def note_pose(self, pos: np.ndarray):
        """An insert without points (pose bookkeeping of tsdf.cpp:46-61 only); pos: float32[3], C-contiguous.  The cheap path
        the frame-sharded map takes on the ranks that do not own a frame."""
        s = self._lib.chad_map_insert_device(self._h, None, 0, pos.ctypes.data)
        if s:
            _lib.check(s, "TSDFMap.note_pose")

    def would_roll_over(self, pos: np.ndarray) -> bool:
        """tsdf.cpp:50-52 as a query: is pos (float32[3]) more than 5 m from the active submap's first pose?"""
        s = self._lib.chad_map_would_roll_over(self._h, pos.ctypes.data, self._yes_ref)
        if s:
            _lib.check(s, "TSDFMap.would_roll_over")
        return self._yes.value != 0

    def save(self, filename: str):
        """TSDFMap::save -- tsdf.cpp:76-86: close the active submap, reconstruct the FIRST submap and write the mesh (binary PLY)."""
        self.close_submap()
        self.finalize()  # Submap::finalize of every submap closed so far (tsdf.cpp:53,79)
        if self.closed_submaps == 0:
            raise RuntimeError("TSDFMap.save: nothing was inserted")
        _lib.check(self._lib.chad_map_save_mesh(self._h, 0, filename.encode()), "TSDFMap.save")

    # ---- the read-back API the reference sketches (tsdf.hpp:120-161) ---------------------------------------
    def iter_leaves(self, submap: int, chunk: int = 1 << 16):
        """Leaves of a closed submap in ascending Morton order, read through the DAG: yields structured arrays with fields
        x, y, z, signed_distance, weight (chad_leaf), `chunk` leaves at a time."""
        it = LeafIterator(self, submap)
        try:
            while it.valid:
                yield it.read(chunk)
        finally:
            it.close()

    def query_sd(self, points, trilinear: bool = True):
        """(sd f32 [n], weight u32 [n]) of the ACTIVE submap at the given points (numpy, host): the containing voxel's leaf or the
        trilinear interpolation of the eight leaves around the point; NaN / 0 where a leaf is missing."""
        pts = _np_f32(points)
        sd = np.empty(pts.shape[0], dtype=np.float32)
        w = np.empty(pts.shape[0], dtype=np.uint32)
        _lib.check(self._lib.chad_map_query_sd(self._h, pts.ctypes.data, pts.shape[0], _lib.QUERY_TRILINEAR if trilinear else _lib.QUERY_LEAF, sd.ctypes.data, w.ctypes.data),
                   "TSDFMap.query_sd")
        return sd, w

    def raycast(self, origins, directions, max_range: float):
        """Distance along each ray (unit directions) to the first sd = 0 crossing of the active submap, NaN if none within max_range."""
        o, d = _np_f32(origins), _np_f32(directions)
        assert o.shape == d.shape
        t = np.empty(o.shape[0], dtype=np.float32)
        _lib.check(self._lib.chad_map_raycast(self._h, o.ctypes.data, d.ctypes.data, o.shape[0], max_range, t.ctypes.data), "TSDFMap.raycast")
        return t

    def submap_positions(self, submap: int = -1) -> np.ndarray:
        """The sensor poses of a submap's frames (Submap::positions, submap.hpp:110), float32 [frames, 3]."""
        n = C.c_size_t()
        _lib.check(self._lib.chad_map_submap_positions(self._h, submap, None, 0, C.byref(n)), "TSDFMap.submap_positions")
        out = np.empty((n.value, 3), dtype=np.float32)
        _lib.check(self._lib.chad_map_submap_positions(self._h, submap, out.ctypes.data, n.value, C.byref(n)), "TSDFMap.submap_positions")
        return out

    def set_submap_pose(self, submap: int, rotation, translation):
        """Loop closure: re-pose a closed submap (x' = R x + t); its iterator positions and mesh vertices come out corrected."""
        r = np.ascontiguousarray(rotation, dtype=np.float32).reshape(9)
        t = np.ascontiguousarray(translation, dtype=np.float32).reshape(3)
        _lib.check(self._lib.chad_map_set_submap_pose(self._h, submap, r.ctypes.data, t.ctypes.data), "TSDFMap.set_submap_pose")

    def mesh_normals(self, submap: int = 0):
        """Unit normal per vertex of mesh(submap) (f32 [V,3]): the normalised sum of the adjacent faces' unit normals (lvr2.cpp:296-297)."""
        nv = C.c_size_t()
        _lib.check(self._lib.chad_map_mesh_normals(self._h, submap, None, 0, C.byref(nv)), "TSDFMap.mesh_normals")
        n = np.empty((nv.value, 3), dtype=np.float32)
        _lib.check(self._lib.chad_map_mesh_normals(self._h, submap, n.ctypes.data, nv.value, C.byref(nv)), "TSDFMap.mesh_normals")
        return n

    def mesh(self, submap: int = 0):
        """(vertices f32 [V,3], triangles u32 [T,3]) of the sd = 0 surface of a submap (closed: >= 0; active: -1)."""
        nv, nt = C.c_size_t(), C.c_size_t()
        _lib.check(self._lib.chad_map_mesh(self._h, submap, None, 0, None, 0, C.byref(nv), C.byref(nt)), "TSDFMap.mesh")
        v = np.empty((nv.value, 3), dtype=np.float32)
        t = np.empty((nt.value, 3), dtype=np.uint32)
        _lib.check(self._lib.chad_map_mesh(self._h, submap, v.ctypes.data, nv.value, t.ctypes.data, nt.value, C.byref(nv), C.byref(nt)), "TSDFMap.mesh")
        return v, t

    # ---- extensions ---------------------------------------------------------------------------------
    def mesh_cells(self, submap: int = 0):
        """The grid the mesh is cut from (ChadGrid, lvr2.cpp:15-133): (codes u64 [C] ascending, corners f32 [C, 8]) of the complete cells."""
        n = C.c_size_t()
        _lib.check(self._lib.chad_map_mesh_cells(self._h, submap, None, None, 0, C.byref(n)), "TSDFMap.mesh_cells")
        codes = np.empty(n.value, dtype=np.uint64)
        corners = np.empty((n.value, 8), dtype=np.float32)
        _lib.check(self._lib.chad_map_mesh_cells(self._h, submap, codes.ctypes.data, corners.ctypes.data, n.value, C.byref(n)), "TSDFMap.mesh_cells")
        return codes, corners

    def sync(self):
        _lib.check(self._lib.chad_map_sync(self._h), "TSDFMap.sync")

    def reset(self):
        _lib.check(self._lib.chad_map_reset(self._h), "TSDFMap.reset")

    def close_submap(self):
        _lib.check(self._lib.chad_map_close_submap(self._h), "TSDFMap.close_submap")

    def stats(self) -> dict:
        st = _lib.Stats()
        _lib.check(self._lib.chad_map_stats(self._h, C.byref(st)), "TSDFMap.stats")
        return {name: getattr(st, name) for name, _ in _lib.Stats._fields_}

    def host_times(self) -> dict:
        """Host time of the insert calls since the last call: per insert in microseconds -- all of it, the wait for a frame slot (the device's
        pace), the H2D copy of host frames, and the rest (the host's own work: bookkeeping and launches)."""
        t = _lib.HostTimes()
        _lib.check(self._lib.chad_map_host_times(self._h, C.byref(t)), "TSDFMap.host_times")
        n = max(1, int(t.inserts))
        return {"inserts": int(t.inserts), "us_per_insert": t.insert_s / n * 1e6, "us_wait": t.wait_s / n * 1e6, "us_copy": t.copy_s / n * 1e6,
                "us_host_work": (t.insert_s - t.wait_s - t.copy_s) / n * 1e6}

    def pipeline_times(self) -> dict:
        """Where the roll-overs' time went and what was allocated since the last call (chad_pipeline_times; the counters restart)."""
        t = _lib.PipelineTimes()
        t.struct_size = C.sizeof(_lib.PipelineTimes)
        _lib.check(self._lib.chad_map_pipeline_times(self._h, C.byref(t)), "TSDFMap.pipeline_times")
        return {name: getattr(t, name) for name, _ in _lib.PipelineTimes._fields_ if name != "struct_size"}

    def kernel_times(self) -> dict:
        """Mean duration (ms) and launch count of every kernel of the insert pipeline since the last call (time_kernels=True)."""
        kt = _lib.KernelTimes()
        _lib.check(self._lib.chad_map_kernel_times(self._h, C.byref(kt)), "TSDFMap.kernel_times")
        return {self._lib.chad_kernel_name(i).decode(): (float(kt.ms[i]), int(kt.launches[i])) for i in range(_lib.KERNEL_COUNT) if kt.launches[i]}

    @property
    def closed_submaps(self) -> int:
        return int(self.stats()["closed_submaps"])

    def export_leaves(self, submap: int = -1):
        """(mc u64, sd f32, w u32) of all voxels with weight > 0, sorted by Morton code."""
        n = C.c_size_t()
        _lib.check(self._lib.chad_map_leaf_count(self._h, submap, C.byref(n)), "TSDFMap.export_leaves")
        mc = np.empty(n.value, dtype=np.uint64)
        sd = np.empty(n.value, dtype=np.float32)
        w = np.empty(n.value, dtype=np.uint32)
        got = C.c_size_t()
        _lib.check(self._lib.chad_map_export_leaves(self._h, submap, mc.ctypes.data, sd.ctypes.data, w.ctypes.data, n.value, C.byref(got)), "TSDFMap.export_leaves")
        assert got.value == n.value
        return mc, sd, w

    def last_frame(self):
        """(order, codes, normals) of the last insert in the device's processing order (parity tests)."""
        n = C.c_size_t()
        _lib.check(self._lib.chad_map_last_frame(self._h, None, None, None, 0, C.byref(n)), "TSDFMap.last_frame")
        order = np.empty(n.value, dtype=np.uint32)
        mc = np.empty(n.value, dtype=np.uint64)
        nrm = np.empty((n.value, 3), dtype=np.float32)
        _lib.check(self._lib.chad_map_last_frame(self._h, order.ctypes.data, mc.ctypes.data, nrm.ctypes.data, n.value, C.byref(n)), "TSDFMap.last_frame")
        return order, mc, nrm

    def block_count(self, submap: int = -1) -> int:
        n = C.c_size_t()
        _lib.check(self._lib.chad_map_block_count(self._h, submap, C.byref(n)), "TSDFMap.block_count")
        return n.value

    def export_blocks(self, submap: int = -1, ordered: bool = True):
        """Per-block partial sums as torch CUDA tensors (keys int64 [B], sum f32 [B,64], cnt int32 [B,64]), sorted by key
        unless ordered=False (the multi-GPU merge sorts the union of all ranks' keys itself)."""
        import torch

        b = self.block_count(submap)
        dev = torch.device("cuda", torch.cuda.current_device())
        keys = torch.empty(b, dtype=torch.int64, device=dev)
        s = torch.empty((b, 64), dtype=torch.float32, device=dev)
        c = torch.empty((b, 64), dtype=torch.int32, device=dev)
        call = self._lib.chad_map_export_blocks_device if ordered else self._lib.chad_map_export_blocks_device_unordered
        _lib.check(call(self._h, submap, keys.data_ptr(), s.data_ptr(), c.data_ptr(), b), "TSDFMap.export_blocks")
        return keys, s, c

    def close_submap_with_blocks(self, keys, s, c):
        """Archive (merged) blocks as the closed submap and start a fresh active one."""
        import torch

        torch.cuda.synchronize()
        b = int(keys.numel())
        _lib.check(self._lib.chad_map_close_submap_with_blocks_device(self._h, keys.data_ptr() if b else None, s.data_ptr() if b else None,
                                                                      c.data_ptr() if b else None, b), "TSDFMap.close_submap_with_blocks")

    # ---- Submap::finalize / NodeLevels (submap.hpp:10-106, levels.hpp:141-199) ---------------------------
    def finalize(self):
        """Fold every closed submap that is not in the DAG yet into it (lazy; the dag_* calls do it themselves)."""
        _lib.check(self._lib.chad_map_finalize(self._h), "TSDFMap.finalize")

    def dag_sizes(self):
        """(entries u64 [21], words u64 [21]) of the DAG's levels as they stand: what dag_reserve() of a later map (or of this one after a
        reset) takes."""
        st = self.dag_stats()
        words = np.zeros(21, dtype=np.uint64)
        n = C.c_size_t()
        for level in range(21):
            _lib.check(self._lib.chad_map_dag_level_words(self._h, level, C.byref(n)), "TSDFMap.dag_sizes")
            words[level] = n.value
        return np.asarray(st["uniques"], dtype=np.uint64), words

    def dag_reserve(self, entries, words):
        """Make the DAG's levels their expected final size from the start (chad_map_dag_reserve)."""
        e = np.ascontiguousarray(entries, dtype=np.uint64).reshape(21)
        w = np.ascontiguousarray(words, dtype=np.uint64).reshape(21)
        _lib.check(self._lib.chad_map_dag_reserve(self._h, e.ctypes.data, w.ctypes.data), "TSDFMap.dag_reserve")

    def dag_roots(self, submap: int):
        """(root_addr_tsdf, root_addr_weight) of a closed submap -- submap.hpp:108-109."""
        rt, rw = C.c_uint32(), C.c_uint32()
        _lib.check(self._lib.chad_map_dag_roots(self._h, submap, C.byref(rt), C.byref(rw)), "TSDFMap.dag_roots")
        return rt.value, rw.value

    def dag_level(self, level: int) -> np.ndarray:
        """Raw words of NodeLevels::_nodes[level] (uint32, level 0..19) or of the leaf clusters (uint64, level 20)."""
        n = C.c_size_t()
        _lib.check(self._lib.chad_map_dag_level_words(self._h, level, C.byref(n)), "TSDFMap.dag_level")
        out = np.empty(n.value, dtype=np.uint64 if level == _lib.DAG_CLUSTER_LEVEL else np.uint32)
        _lib.check(self._lib.chad_map_dag_level_copy(self._h, level, out.ctypes.data, n.value), "TSDFMap.dag_level")
        return out

    def dag_stats(self) -> dict:
        st = _lib.DagStats()
        st.struct_size = C.sizeof(_lib.DagStats)
        _lib.check(self._lib.chad_map_dag_stats(self._h, C.byref(st)), "TSDFMap.dag_stats")
        return {"submaps": int(st.submaps), "uniques": list(st.uniques), "dupes": list(st.dupes), "ms_device": st.ms_device, "ms_host": st.ms_host, "device_bytes": int(st.device_bytes)}

    def dag_leaves(self, submap: int):
        """(mc u64, sd f32) read back through the DAG (lvr2.cpp:30-90), sd de-quantised from 8 bits, ascending by code."""
        n = C.c_size_t()
        _lib.check(self._lib.chad_map_dag_leaves(self._h, submap, None, None, 0, C.byref(n)), "TSDFMap.dag_leaves")
        mc = np.empty(n.value, dtype=np.uint64)
        sd = np.empty(n.value, dtype=np.float32)
        _lib.check(self._lib.chad_map_dag_leaves(self._h, submap, mc.ctypes.data, sd.ctypes.data, n.value, C.byref(n)), "TSDFMap.dag_leaves")
        return mc, sd

    def export_clusters(self, submap: int = -1, weights: bool = True):
        """Device half of finalize on its own: (block keys [B] ascending, tsdf words [B,8], weight words [B,8] or None)."""
        b = C.c_size_t()
        _lib.check(self._lib.chad_map_export_clusters(self._h, submap, None, None, None, 0, C.byref(b)), "TSDFMap.export_clusters")
        keys = np.empty(b.value, dtype=np.uint64)
        tsdf = np.empty((b.value, 8), dtype=np.uint64)
        wts = np.empty((b.value, 8), dtype=np.uint64) if weights else None
        _lib.check(self._lib.chad_map_export_clusters(self._h, submap, keys.ctypes.data, tsdf.ctypes.data, wts.ctypes.data if weights else None, b.value, C.byref(b)),
                   "TSDFMap.export_clusters")
        return keys, tsdf, wts


LEAF_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("signed_distance", "<f4"), ("weight", "<u4"), ("reserved", "<u4")])


class LeafIterator:
    """chad::TSDFMap::iterator (tsdf.hpp:126-151): a position among the leaves of one closed submap."""

    def __init__(self, tsdf_map: TSDFMap, submap: int):
        self._lib = tsdf_map._lib
        self._h = C.c_void_p()
        _lib.check(self._lib.chad_map_iter_begin(tsdf_map._h, submap, C.byref(self._h)), "LeafIterator")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.chad_iter_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def valid(self) -> bool:
        return bool(self._lib.chad_iter_valid(self._h))

    def seek(self, last: bool = False):
        _lib.check(self._lib.chad_iter_seek(self._h, int(last)), "LeafIterator.seek")

    def advance(self, n: int = 1) -> int:
        moved = C.c_int64()
        _lib.check(self._lib.chad_iter_advance(self._h, n, C.byref(moved)), "LeafIterator.advance")
        return moved.value

    def get(self):
        leaf = _lib.Leaf()
        _lib.check(self._lib.chad_iter_get(self._h, C.byref(leaf)), "LeafIterator.get")
        return (leaf.x, leaf.y, leaf.z), leaf.signed_distance, leaf.weight

    def read(self, capacity: int) -> np.ndarray:
        out = np.empty(capacity, dtype=LEAF_DTYPE)
        n = C.c_size_t()
        _lib.check(self._lib.chad_iter_read(self._h, out.ctypes.data, capacity, C.byref(n)), "LeafIterator.read")
        return out[: n.value]


# ---- stage entry points (parity tests call the device through these) ---------------------------------
def morton_codes(points, sdf_res: float) -> np.ndarray:
    lib = _lib.load()
    pts = _np_f32(points)
    out = np.empty(pts.shape[0], dtype=np.uint64)
    _lib.check(lib.chad_morton_codes(pts.ctypes.data, pts.shape[0], sdf_res, out.ctypes.data), "morton_codes")
    return out


def walk_rays(points, position, sdf_res: float, sdf_trunc: float, stride: int = 32):
    lib = _lib.load()
    pts = _np_f32(points)
    pos = _np_f32(position).reshape(3)
    vox = np.zeros((pts.shape[0], stride, 3), dtype=np.int32)
    cnt = np.zeros(pts.shape[0], dtype=np.uint32)
    _lib.check(lib.chad_walk_rays(pts.ctypes.data, pts.shape[0], pos.ctypes.data, sdf_res, sdf_trunc, vox.ctypes.data, cnt.ctypes.data, stride), "walk_rays")
    return vox, cnt


def sort_pairs(keys, key_bits: int = 64):
    """Stable ascending radix sort of u64 keys on the device: (sorted keys, input positions)."""
    lib = _lib.load()
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    out = np.empty_like(k)
    order = np.empty(k.shape[0], dtype=np.uint32)
    _lib.check(lib.chad_sort_pairs(k.ctypes.data, k.shape[0], key_bits, out.ctypes.data, order.ctypes.data), "sort_pairs")
    return out, order


def decode_morton(mc: np.ndarray) -> np.ndarray:
    """MortonCode::decode (morton.hpp:29-37) for an array of codes -> int32 [n,3] (host helper for save())."""
    def compact(v):
        v = v & np.uint64(0x1249249249249249)
        v = (v | (v >> np.uint64(2))) & np.uint64(0x10C30C30C30C30C3)
        v = (v | (v >> np.uint64(4))) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v >> np.uint64(8))) & np.uint64(0x001F0000FF0000FF)
        v = (v | (v >> np.uint64(16))) & np.uint64(0x001F00000000FFFF)
        v = (v | (v >> np.uint64(32))) & np.uint64(0x1FFFFF)
        return (v.astype(np.int64) - (1 << 20)).astype(np.int32)

    mc = np.asarray(mc, dtype=np.uint64)
    return np.stack([compact(mc), compact(mc >> np.uint64(1)), compact(mc >> np.uint64(2))], axis=-1)

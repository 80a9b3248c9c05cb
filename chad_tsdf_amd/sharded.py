"""Frame-sharded TSDF map: one process per GPU, whole frames dealt round-robin, one exchange per submap.

Why this shards: within a frame, every stage of TSDFMap::insert depends only on that frame's
points and pose (tsdf.cpp:64-70); the only cross-frame state is the per-voxel mean and count, and the reference's
update is a mean of unit-weight observations (octree.hpp:161-163), so per-GPU partial (sum, count) bricks combine
exactly by addition.  Frame k of the stream goes to rank k % world_size; the roll-over decision (tsdf.cpp:46-61)
depends only on the pose sequence, so every rank takes it identically without communication.

All of it lives behind the C ABI (include/chad_hip.h, "Multi-GPU"): chad_map_attach_comm makes insert() frame-sharded and the
exchange at a submap close runs in the library, on 64-bit integer sums, so N ranks build bit for bit what one GPU builds:
  * a map that keeps its closed submaps' dense archives: all-gather of block keys, sorted union, all-reduce of dense [U x 64]
    sums and counts -- every rank ends up with the merged submap (merge_kernels.hpp);
  * a map under CHAD_FLAG_DAG_ONLY (long trajectories, BASELINE config 5): the SPARSE exchange -- keys, voxel masks and one word
    per touched voxel to the owner of a Morton key range, each range's cluster words to rank 0, Submap::finalize once, there
    (exchange_kernels.hpp).
This module only builds the communicator (`Comm`: the RCCL id travels over torch.distributed; with the gloo backend the library's
collectives are served by torch.distributed callbacks on host buffers, which is how several ranks are rehearsed on one GPU) and
keeps the SPMD wrapper.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


class Comm:
    """A chad_comm over the ranks of a torch.distributed process group.

    backend nccl: the library's own RCCL communicator (rank 0's ncclUniqueId is broadcast through the process group);
    backend gloo (or anything else): the library calls back into torch.distributed with host buffers."""

    def __init__(self, group=None):
        import ctypes as C

        from . import _lib

        self._lib = _lib.load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.group = group
        self.handle = C.c_void_p()
        backend = dist.get_backend(group)
        if backend == "nccl":
            uid = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
            if self.rank == 0:
                _lib.check(self._lib.chad_comm_unique_id(uid.numpy().ctypes.data), "Comm")
            uid = uid.cuda()
            dist.broadcast(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = uid.cpu().numpy()
            torch.cuda.synchronize()
            _lib.check(self._lib.chad_comm_create_rccl(uid.ctypes.data, self.rank, self.world, C.byref(self.handle)), "Comm")
            self.transport = "rccl"
        else:
            def view(ptr, nbytes, dtype):
                return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)).view(dtype))

            def all_gather(_user, send, recv, nbytes):
                try:
                    dist.all_gather_into_tensor(view(recv, nbytes * self.world, np.uint8), view(send, nbytes, np.uint8), group=group)
                    return 0
                except Exception:  # noqa: BLE001  (reported by the library as a failed exchange)
                    import traceback

                    traceback.print_exc()
                    return 1

            def all_reduce(_user, buf, count, dtype):
                try:   # wrapping integer sums: the signed view adds the same bits
                    t = view(buf, count * 8, np.int64) if dtype == _lib.REDUCE_I64 else view(buf, count * 4, np.int32)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                    return 0
                except Exception:  # noqa: BLE001
                    import traceback

                    traceback.print_exc()
                    return 1

            self._callbacks = (_lib.ALLGATHER_FN(all_gather), _lib.ALLREDUCE_FN(all_reduce))   # kept alive with the communicator
            _lib.check(self._lib.chad_comm_create_callbacks(self.rank, self.world, C.cast(self._callbacks[0], C.c_void_p), C.cast(self._callbacks[1], C.c_void_p),
                                                            None, C.byref(self.handle)), "Comm")
            self.transport = "callbacks/" + backend

    def close(self):
        if getattr(self, "handle", None):
            self._lib.chad_comm_destroy(self.handle)
            self.handle = None

    __del__ = close


class ShardedTSDFMap:
    """SPMD wrapper: every rank calls insert() for every frame, in the same order, with the same position.  Points are only needed on
    the rank that owns the frame (frame index % world size); the other ranks may pass None.  `local`: this rank's TSDFMap (made here
    from map_kwargs by default).  Ownership, pose bookkeeping and the exchange at a roll-over all happen in the library."""

    def __init__(self, sdf_res: float = 0.05, sdf_trunc: float = 0.1, *, local=None, group=None, comm=None, **map_kwargs):
        if local is None:
            from .tsdf import TSDFMap

            local = TSDFMap(sdf_res, sdf_trunc, **map_kwargs)
        self.local = local
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.comm = comm if comm is not None else (Comm(group) if self.world > 1 else None)   # (a Comm may serve several maps, one after the other)
        local.attach_comm(self.comm)
        self._open = False   # the active submap holds a pose

    def owner(self, frame: int) -> int:
        return frame % self.world

    def insert(self, points, position):
        pos = np.ascontiguousarray(position, dtype=np.float32).reshape(3)
        if points is None:
            self.local.note_pose(pos)
        else:
            self.local.insert(points, pos)
        self._open = True

    def merge_and_close(self):
        """The exchange step (collective): merge the active submap across the ranks and close it."""
        self.local.merge_and_close()
        self._open = False

    def save(self, filename: str):
        """TSDFMap::save (tsdf.cpp:76-86): merge + close the active submap, rank 0 writes the first submap."""
        if self._open:
            self.merge_and_close()
        if self.rank == 0:
            self.local.save(filename)

    @property
    def closed_submaps(self) -> int:
        return self.local.closed_submaps

    def export_leaves(self, submap: int):
        """Leaves of a CLOSED (merged) submap; identical on every rank (maps that keep their archives)."""
        if submap < 0:
            raise ValueError("the active submap is spread over the ranks; call merge_and_close() first")
        return self.local.export_leaves(submap)

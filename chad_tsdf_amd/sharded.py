"""Frame-sharded TSDF map: one process per GPU, whole frames dealt round-robin, one exchange per submap.

Why this shards: within a frame, every stage of TSDFMap::insert depends only on that frame's
points and pose (tsdf.cpp:64-70); the only cross-frame state is the per-voxel mean and count, and the reference's
update is a mean of unit-weight observations (octree.hpp:161-163), so per-GPU partial (sum, count) bricks combine
exactly by addition.  Frame k of the stream goes to rank k % world_size; the roll-over decision (tsdf.cpp:46-61)
depends only on the pose sequence, so every rank takes it identically without communication.  The exchange
happens when a submap closes:

  1. all_gather of each rank's sorted block keys (counts first, then keys padded to the longest list);
  2. every rank forms the same sorted union and scatters its bricks into dense [B_union, 64] buffers;
  3. all_reduce(SUM) of the f32 sums and of the i32 counts (backend "nccl" = RCCL over xGMI on ROCm);
  4. every rank archives the merged bricks as the closed submap (sd = sum / count at read-back).

With the HIP map all of this lives behind the C ABI (include/chad_hip.h, "Multi-GPU"): chad_map_attach_comm makes insert()
frame-sharded and the exchange runs in the library -- RCCL all-gather / all-reduce of 64-bit integer sums and counts, so N ranks
archive bit for bit what one GPU archives.  This module only builds the communicator (`Comm`: the RCCL id travels over
torch.distributed; with the gloo backend the library's collectives are served by torch.distributed callbacks on host buffers,
which is how several ranks are rehearsed on one GPU) and keeps the SPMD wrapper.  `merge_blocks` and the Python protocol below
remain for duck-typed stand-in maps (the CPU tests of the sharding logic, tests/test_distributed_gloo.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

SUBMAP_DISTANCE = np.float32(5.0)  # tsdf.cpp:52


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


def merge_blocks(keys: torch.Tensor, sums: torch.Tensor, cnts: torch.Tensor, group=None):
    """Combine per-rank partial bricks.  keys int64 [B] (distinct, in any order), sums f32 [B,64], cnts i32 [B,64].

    Returns (union keys [U], summed sums [U,64], summed counts [U,64]), identical on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return keys, sums, cnts
    home = keys.device
    if keys.is_cuda and dist.get_backend(group) == "gloo":  # rehearsal of the GPU path over gloo: collectives on host copies
        merged = merge_blocks(keys.cpu(), sums.cpu(), cnts.cpu(), group)
        return tuple(t.to(home) for t in merged)
    dev = keys.device
    n_local = torch.tensor([keys.numel()], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    longest = max(max(counts), 1)
    padded = torch.full((longest,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    padded[: keys.numel()] = keys
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    union = torch.unique(torch.cat([g[:c] for g, c in zip(gathered, counts)]))  # sorted
    dense_sum = torch.zeros((union.numel(), 64), dtype=torch.float32, device=dev)
    dense_cnt = torch.zeros((union.numel(), 64), dtype=torch.int32, device=dev)
    if keys.numel():
        at = torch.searchsorted(union, keys)
        dense_sum[at] = sums
        dense_cnt[at] = cnts
    if union.numel():
        dist.all_reduce(dense_sum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(dense_cnt, op=dist.ReduceOp.SUM, group=group)
    return union, dense_sum, dense_cnt


class ShardedTSDFMap:
    """SPMD wrapper: every rank calls insert() for every frame, in the same order, with the same position.

    `local` is this rank's single-device map: chad_tsdf_amd.TSDFMap on a GPU (the default), or any object with
    insert / export_blocks / close_submap_with_blocks / export_leaves (the gloo tests pass a CPU stand-in).
    Points are only needed on the rank that owns the frame (frame index % world size); other ranks may pass None.
    """

    def __init__(self, sdf_res: float = 0.05, sdf_trunc: float = 0.1, *, local=None, group=None, comm=None, **map_kwargs):
        if local is None:
            from .tsdf import TSDFMap

            local = TSDFMap(sdf_res, sdf_trunc, **map_kwargs)
        self.local = local
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._first_pose = None  # first pose of the active submap (positions.front(), tsdf.cpp:51)
        self._frame = 0          # frames seen in the stream
        self._empty = np.zeros((0, 3), dtype=np.float32)
        self._fast = hasattr(local, "would_roll_over") and hasattr(local, "note_pose")  # the HIP map; stand-ins take the numpy path
        # the HIP map shards inside the library: attach a communicator and every insert / roll-over / close does the right thing there
        self._native = hasattr(local, "attach_comm")
        self.comm = None
        if self._native:
            self.comm = comm if comm is not None else (Comm(group) if self.world > 1 else None)   # (a Comm may serve several maps, one after the other)
            local.attach_comm(self.comm)

    def owner(self, frame: int) -> int:
        return frame % self.world

    def _rolls_over(self, pos: np.ndarray) -> bool:
        # tsdf.cpp:52 in f32, same operation order as glm::distance (and as the C library's host code)
        d = pos - self._first_pose
        return bool(np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > SUBMAP_DISTANCE)

    def insert(self, points, position):
        pos = np.ascontiguousarray(position, dtype=np.float32).reshape(3)
        if self._native:   # ownership (frame k -> rank k % world), pose bookkeeping and the exchange at a roll-over all happen in the library
            if points is None:
                self.local.note_pose(pos)
            else:
                self.local.insert(points, pos)
            self._first_pose = pos
            self._frame += 1
            return
        # The roll-over rule, evaluated identically on every rank before the frame.  A rank that does not own the frame
        # spends about 3 us here (one query + one pose-only insert, both plain C calls): at 8 ranks that is what bounds the
        # strong scaling of a stream of small frames, so the path avoids numpy arithmetic where the local map offers the calls.
        if self._fast:
            if self.local.would_roll_over(pos):   # false while the active submap holds no pose
                self.merge_and_close()      # collective; leaves every local map with an empty active submap
            self._first_pose = pos          # (only "a submap is open" matters on this path; the library keeps the poses)
            if self._frame % self.world == self.rank:
                self.local.insert(points, pos)
            else:
                self.local.note_pose(pos)
            self._frame += 1
            return
        if self._first_pose is None:
            self._first_pose = pos.copy()
        elif self._rolls_over(pos):
            self.merge_and_close()          # collective; leaves every local map with an empty active submap
            self._first_pose = pos.copy()   # the frame that triggers the roll-over opens the new submap (tsdf.cpp:55-57)
        mine = self.owner(self._frame) == self.rank
        # ranks that do not own the frame still record its pose (an empty insert runs only the bookkeeping of
        # tsdf.cpp:46-61), so all local maps agree on the submap's first pose
        self.local.insert(points if mine else self._empty, pos)
        self._frame += 1

    def merge_and_close(self):
        """The exchange step: merge the active submap across ranks and archive it on every rank."""
        if self._native:
            self.local.merge_and_close()
            self._first_pose = None
            return
        try:
            keys, sums, cnts = self.local.export_blocks(-1, ordered=False)   # the merge orders the union of all ranks' keys
        except TypeError:  # a stand-in map without the option
            keys, sums, cnts = self.local.export_blocks(-1)
        keys, sums, cnts = merge_blocks(keys, sums, cnts, self.group)
        self.local.close_submap_with_blocks(keys, sums, cnts)
        self._first_pose = None

    def save(self, filename: str):
        """TSDFMap::save (tsdf.cpp:76-86): merge + close the active submap, rank 0 writes the first submap."""
        if self._first_pose is not None:
            self.merge_and_close()
        if self.rank == 0:
            self.local.save(filename)

    @property
    def closed_submaps(self) -> int:
        return self.local.closed_submaps

    def export_leaves(self, submap: int):
        """Leaves of a CLOSED (merged) submap; identical on every rank."""
        if submap < 0:
            raise ValueError("the active submap is spread over the ranks; call merge_and_close() first")
        return self.local.export_leaves(submap)

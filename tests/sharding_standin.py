"""TEST INFRASTRUCTURE: the frame-sharding PROTOCOL in plain Python over duck-typed local maps (the CPU tests pass an oracle-backed
stand-in), and the dense merge it uses.  The product shards inside the library (chad_map_attach_comm; chad_tsdf_amd/sharded.py is a thin
wrapper); this file restates the protocol -- frame k to rank k % world, the roll-over rule evaluated identically on every rank
(tsdf.cpp:46-61), one merge per submap -- so that it can be exercised over gloo without a GPU."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

SUBMAP_DISTANCE = np.float32(5.0)  # tsdf.cpp:52


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


class StandinShardedMap:
    """Every rank calls insert() for every frame, in the same order, with the same position; `local` has insert / export_blocks /
    close_submap_with_blocks / export_leaves."""

    def __init__(self, local, group=None):
        self.local, self.group = local, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._first_pose = None  # first pose of the active submap (positions.front(), tsdf.cpp:51)
        self._frame = 0
        self._empty = np.zeros((0, 3), dtype=np.float32)

    def _rolls_over(self, pos: np.ndarray) -> bool:
        # tsdf.cpp:52 in f32, same operation order as glm::distance (and as the C library's host code)
        d = pos - self._first_pose
        return bool(np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > SUBMAP_DISTANCE)

    def insert(self, points, position):
        pos = np.ascontiguousarray(position, dtype=np.float32).reshape(3)
        if self._first_pose is None:
            self._first_pose = pos.copy()
        elif self._rolls_over(pos):
            self.merge_and_close()          # collective; leaves every local map with an empty active submap
            self._first_pose = pos.copy()   # the frame that triggers the roll-over opens the new submap (tsdf.cpp:55-57)
        mine = self._frame % self.world == self.rank
        # ranks that do not own the frame still record its pose (an empty insert runs only the bookkeeping of tsdf.cpp:46-61)
        self.local.insert(points if mine else self._empty, pos)
        self._frame += 1

    def merge_and_close(self):
        keys, sums, cnts = self.local.export_blocks(-1)
        keys, sums, cnts = merge_blocks(keys, sums, cnts, self.group)
        self.local.close_submap_with_blocks(keys, sums, cnts)
        self._first_pose = None

    @property
    def closed_submaps(self) -> int:
        return self.local.closed_submaps

    def export_leaves(self, submap: int):
        return self.local.export_leaves(submap)

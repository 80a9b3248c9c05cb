"""The N > 1 path on CPU: two ranks (gloo), frames dealt round-robin, partial bricks merged at submap close.

The per-rank single-device map is a CPU stand-in built on the oracle (this is test infrastructure: the product's
ShardedTSDFMap only sees the duck-typed `local` object); what is under test is the sharding logic and the
exchange step of chad_tsdf_amd/sharded.py -- the same code that runs over RCCL on GPUs."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleLocalMap:
    """insert / export_blocks / close_submap_with_blocks / export_leaves over the oracle, with CPU tensors."""

    def __init__(self, res, trunc):
        from oracle import pyoracle

        self.res, self.trunc = res, trunc
        self.m = pyoracle.OracleMap(res, trunc)
        self.closed = []

    def insert(self, points, position):
        self.m.insert(points, position)
        assert self.m.closed_submaps == 0, "the sharded wrapper must pre-empt the local roll-over"

    def export_blocks(self, submap=-1):
        mc, sd, w = self.m.export_leaves(-1)
        keys, inv = np.unique(mc & ~np.uint64(63), return_inverse=True)
        s = np.zeros((len(keys), 64), np.float32)
        c = np.zeros((len(keys), 64), np.int32)
        vox = (mc & np.uint64(63)).astype(np.int64)
        s[inv, vox] = sd * w.astype(np.float32)
        c[inv, vox] = w.astype(np.int32)
        return torch.from_numpy(keys.astype(np.int64)), torch.from_numpy(s), torch.from_numpy(c)

    def close_submap_with_blocks(self, keys, s, c):
        self.closed.append((keys.clone(), s.clone(), c.clone()))
        self.m.reset()

    @property
    def closed_submaps(self):
        return len(self.closed)

    def export_leaves(self, submap):
        keys, s, c = self.closed[submap]
        keys, s, c = keys.numpy().astype(np.uint64), s.numpy(), c.numpy()
        b, v = np.nonzero(c)
        return keys[b] | v.astype(np.uint64), s[b, v] / c[b, v].astype(np.float32), c[b, v].astype(np.uint32)


def _frames():
    from chad_tsdf_amd import synth

    out = []
    for k in range(14):  # poses 0.6 m apart: roll-over when the 10th pose (5.4 m) arrives; 4 frames in the second submap
        pts, _ = synth.cfg3_frame(7 * k, 3000)
        out.append((pts, np.array([-12.0 + 0.6 * k, 0.1, 0.0], np.float32)))
    return out


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chad_tsdf_amd.sharded import ShardedTSDFMap

    sm = ShardedTSDFMap(0.05, 0.10, local=OracleLocalMap(0.05, 0.10))
    for k, (pts, pos) in enumerate(_frames()):
        sm.insert(pts if k % world == rank else None, pos)  # non-owners do not even get the points
    sm.merge_and_close()
    assert sm.closed_submaps == 2
    for sub in range(2):
        mc, sd, w = sm.export_leaves(sub)
        np.savez(os.path.join(out_dir, f"rank{rank}_sub{sub}.npz"), mc=mc, sd=sd, w=w)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_match_one_map(tmp_path, oracle):
    world = 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ref = oracle.OracleMap(0.05, 0.10)  # the unsharded reference behaviour: one map sees every frame
    for pts, pos in _frames():
        ref.insert(pts, pos)
    assert ref.closed_submaps == 1
    want = [ref.export_leaves(0), ref.export_leaves(-1)]
    for sub in range(2):
        got = [np.load(tmp_path / f"rank{r}_sub{sub}.npz") for r in range(world)]
        for r in range(world):
            assert np.array_equal(got[r]["mc"], want[sub][0]), "leaf set of the merged submap differs"
            assert np.array_equal(got[r]["w"], want[sub][2]), "weights of the merged submap differ"
            err = np.abs(got[r]["sd"] - want[sub][1]) / np.maximum(np.abs(want[sub][1]), 0.10)
            assert err.max() <= 1e-4
        assert np.array_equal(got[0]["sd"], got[1]["sd"]), "ranks disagree after the all-reduce"


def test_merge_blocks_single_rank_is_identity():
    from chad_tsdf_amd.sharded import merge_blocks

    k = torch.tensor([64, 128], dtype=torch.int64)
    s = torch.ones((2, 64))
    c = torch.ones((2, 64), dtype=torch.int32)
    k2, s2, c2 = merge_blocks(k, s, c)
    assert k2 is k and s2 is s and c2 is c

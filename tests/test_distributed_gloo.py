"""The N > 1 path on CPU, two ranks over gloo.

1. The PRODUCT's host arithmetic of the sparse exchange (chad_tsdf_amd/csrc/exchange_plan.hpp through the C ABI's stage entries
   chad_exchange_splitters / chad_exchange_layout -- they need no device): two ranks with different sorted block lists all-gather
   their heads, derive the splitters, their per-range rows and the byte layout of the point-to-point step, run that step the way the
   callback transport does (whole send buffers gathered, slices picked out by the layout's offsets) on slices packed in the
   documented format, and the owners' accumulated voxels must be the sum over both ranks -- key ranges disjoint, ordered, complete.
2. The frame-sharding PROTOCOL restated in Python (tests/sharding_standin.py) over an oracle-backed stand-in map: frames dealt
   round-robin, the roll-over rule taken identically on both ranks, one merge per submap.  (The library's own sharding and both of
   its exchanges run on the GPU: tests/test_gpu_sharded.py, tests/test_gpu_exchange_faults.py.)"""
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
    from tests.sharding_standin import StandinShardedMap

    sm = StandinShardedMap(OracleLocalMap(0.05, 0.10))
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
    from tests.sharding_standin import merge_blocks

    k = torch.tensor([64, 128], dtype=torch.int64)
    s = torch.ones((2, 64))
    c = torch.ones((2, 64), dtype=torch.int32)
    k2, s2, c2 = merge_blocks(k, s, c)
    assert k2 is k and s2 is s and c2 is c


# ---- the product's exchange planning on two gloo ranks ---------------------------------------------------------------------------
S = 64   # kXSamples


def _share(rank):
    """A rank's share of a submap: sorted distinct block keys, a voxel mask per block, one staging word per set voxel (count << 40 | sum)."""
    rng = np.random.default_rng(100 + rank)
    n = 5000 + 3000 * rank
    # block keys: Morton-like 63-bit values with the low six bits clear; the two ranks overlap in about a third of their blocks
    pool = np.sort(np.random.default_rng(7).choice(1 << 40, size=12000, replace=False).astype(np.uint64)) << np.uint64(6)
    keys = np.sort(rng.choice(pool, size=n, replace=False))
    masks = rng.integers(1, 1 << 63, size=n, dtype=np.uint64) & rng.integers(1, 1 << 63, size=n, dtype=np.uint64) | np.uint64(1)
    counts = np.array([bin(int(m)).count("1") for m in masks], dtype=np.int64)
    words = (rng.integers(1, 200, size=int(counts.sum())).astype(np.uint64) << np.uint64(40)) | rng.integers(0, 1 << 20, size=int(counts.sum())).astype(np.uint64)
    return keys, masks, counts, words


def _plan_worker(rank, world, port, out_dir):
    import ctypes as C

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chad_tsdf_amd import _lib

    lib = _lib.load()   # (loads without a GPU; only the two host-side stage entries are called)
    keys, masks, counts, words = _share(rank)
    n = len(keys)
    # stage 0: head = [count | width flag << 32, 64 samples at even distances] (k_xchg_head), all-gathered
    head = np.empty(S + 1, np.uint64)
    head[0] = n
    head[1:] = keys[(np.arange(1, S + 1, dtype=np.int64) * n) // (S + 1)]
    heads = [torch.zeros(S + 1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(heads, torch.from_numpy(head.view(np.int64)))
    heads = np.stack([h.numpy().view(np.uint64) for h in heads])
    split = np.zeros(world - 1, np.uint64)
    _lib.check(lib.chad_exchange_splitters(heads.ctypes.data, world, split.ctypes.data), "splitters")
    # stage 1: my blocks per range (k_xchg_ranges: lower bounds of the splitters), their words; the G x G matrix all-gathered
    first = np.concatenate([[0], np.searchsorted(keys, split, side="left"), [n]]).astype(np.int64)
    prefix = np.concatenate([[0], np.cumsum(counts)])
    row = np.array([[first[d + 1] - first[d], prefix[first[d + 1]] - prefix[first[d]]] for d in range(world)], np.uint64)
    rows = [torch.zeros(2 * world, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(rows, torch.from_numpy(row.reshape(-1).view(np.int64)))
    matrix = np.stack([r.numpy().view(np.uint64) for r in rows]).reshape(world, world, 2)
    wide = np.zeros(world, np.uint32)
    layout = np.zeros(5 * world + 1, np.uint64)
    _lib.check(lib.chad_exchange_layout(np.ascontiguousarray(matrix).ctypes.data, wide.ctypes.data, world, rank, layout.ctypes.data), "layout")
    send_off, send_bytes, recv_off, recv_bytes, peer_off = (layout[:5 * world].reshape(world, 5).astype(np.int64).T)
    longest = int(layout[5 * world])
    # stage 2: the slices, packed as exchange_kernels.hpp documents them: keys | masks | first u32 (padded to 8) | payload
    send = np.zeros(longest, np.uint8)
    for d in range(world):
        a, b = first[d], first[d + 1]
        nb = b - a
        sl = np.concatenate([keys[a:b].view(np.uint8), masks[a:b].view(np.uint8), np.pad((prefix[a:b] - prefix[a]).astype(np.uint32).view(np.uint8), (0, (-4 * nb) % 8)),
                             words[prefix[a]:prefix[b]].view(np.uint8)])
        assert len(sl) == send_bytes[d], "the layout's slice size is the documented format's"
        send[send_off[d]:send_off[d] + len(sl)] = sl
    gathered = [torch.zeros(longest, dtype=torch.uint8) for _ in range(world)]   # the callback transport: whole send buffers gathered
    dist.all_gather(gathered, torch.from_numpy(send))
    # stage 3: what arrives for MY range, accumulated per (key, voxel)
    acc = {}
    lo = 0 if rank == 0 else int(split[rank - 1])
    hi = (1 << 64) if rank == world - 1 else int(split[rank])
    for r in range(world):
        buf = gathered[r].numpy()[peer_off[r]:peer_off[r] + recv_bytes[r]] if r != rank else send[send_off[rank]:send_off[rank] + send_bytes[rank]]
        nb = int(matrix[r, rank, 0])
        k = buf[:8 * nb].view(np.uint64)
        mk = buf[8 * nb:16 * nb].view(np.uint64)
        fi = buf[16 * nb:16 * nb + 4 * nb].view(np.uint32)
        pay = buf[16 * nb + 4 * nb + (-4 * nb) % 8:].view(np.uint64)
        assert len(pay) == int(matrix[r, rank, 1])
        assert nb == 0 or (int(k.min()) >= lo and int(k.max()) < hi), "a key outside the owner's range"
        for i in range(nb):
            j = int(fi[i])
            for v in range(64):
                if (int(mk[i]) >> v) & 1:
                    acc[(int(k[i]), v)] = acc.get((int(k[i]), v), 0) + int(pay[j])
                    j += 1
    np.save(os.path.join(out_dir, f"plan_rank{rank}.npy"), np.array(sorted((k, v, w) for (k, v), w in acc.items()), dtype=np.uint64))
    np.save(os.path.join(out_dir, f"split_rank{rank}.npy"), split)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sparse_exchange_planning_on_two_ranks(tmp_path):
    world = 2
    port = 31500 + os.getpid() % 2000
    mp.spawn(_plan_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    want = {}
    for rank in range(world):
        keys, masks, counts, words = _share(rank)
        j = 0
        for k, mk in zip(keys.tolist(), masks.tolist()):
            for v in range(64):
                if (mk >> v) & 1:
                    want[(k, v)] = want.get((k, v), 0) + int(words[j])
                    j += 1
    got = np.concatenate([np.load(tmp_path / f"plan_rank{r}.npy") for r in range(world)])   # range 0 then range 1: ascending keys
    assert np.array_equal(np.load(tmp_path / "split_rank0.npy"), np.load(tmp_path / "split_rank1.npy")), "the ranks derived different splitters"
    assert len(got) == len(want)
    assert np.all(got[1:, 0] >= got[:-1, 0]), "the ranges, concatenated in rank order, are not in Morton order"
    for k, v, w in got.tolist():
        assert want[(k, v)] == w
    sizes = [len(np.load(tmp_path / f"plan_rank{r}.npy")) for r in range(world)]
    assert min(sizes) > 0.3 * sum(sizes), f"the splitters left the ranges unbalanced: {sizes}"

"""The host arithmetic of the sparse multi-GPU exchange (chad_tsdf_amd/csrc/exchange_plan.hpp) on its own, for 1 to 15 ranks, built with
g++ (and under AddressSanitizer / UBSan by `make asan`): the splitters are ordered and balance the ranks' blocks; every pair of ranks agrees
on the slice they exchange -- a's send size for b is b's receive size from a, and b finds a's slice where a put it; the ranges' way to
rank 0 likewise.  (Two gloo ranks run the same functions through the C ABI and move real slices: test_distributed_gloo.py.)"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import pyoracle   # (only for the sanitizer switch of `make asan`)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "plan_check.cpp")
S = 64


@pytest.fixture(scope="module")
def plan(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("plancheck") / "libplancheck.so")
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] if pyoracle.SANITIZE else ["-O2"]
    subprocess.check_call([shutil.which("g++") or "g++", "-std=c++17", "-Wall", "-Werror", *san, "-shared", "-fPIC", SRC, "-o", out])
    lib = C.CDLL(out)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    lib.plancheck_splitters.argtypes = [vp, u32, vp]
    lib.plancheck_slices.argtypes = [vp, vp, u32, u32, vp]
    lib.plancheck_gather.argtypes = [vp, u32, u32, C.c_int, vp]
    lib.plancheck_slice_bytes.restype = u64
    lib.plancheck_slice_bytes.argtypes = [u64, u64, u32]
    lib.plancheck_range_bytes.restype = u64
    lib.plancheck_range_bytes.argtypes = [u64, u64, C.c_int]
    return lib


def _keys(rng, n):
    return np.sort(rng.choice(1 << 40, size=n, replace=False).astype(np.uint64)) << np.uint64(6)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8, 15])
def test_splitters_and_slice_plans_agree_between_all_ranks(plan, world):
    rng = np.random.default_rng(world)
    counts = [0 if (world > 2 and r == 1) else int(rng.integers(500, 20_000)) for r in range(world)]   # one rank without a block
    keys = [_keys(rng, n) for n in counts]
    heads = np.zeros((world, S + 1), np.uint64)
    for r, k in enumerate(keys):
        heads[r, 0] = len(k) | ((r & 1) << 32)                      # odd ranks send two words per voxel
        heads[r, 1:] = k[(np.arange(1, S + 1) * len(k)) // (S + 1)] if len(k) else np.uint64(0xFFFFFFFFFFFFFFFF)
    split = np.zeros(max(world - 1, 1), np.uint64)
    assert plan.plancheck_splitters(heads.ctypes.data, world, split.ctypes.data) == 0
    split = split[:world - 1]
    assert np.all(split[1:] >= split[:-1]), "splitters out of order"
    # the ranges every rank cuts its sorted list into, the G x G matrix
    matrix = np.zeros((world, world, 2), np.uint64)
    for r, k in enumerate(keys):
        first = np.concatenate([[0], np.searchsorted(k, split, side="left"), [len(k)]]).astype(np.int64)
        voxels = rng.integers(1, 64, size=len(k))
        prefix = np.concatenate([[0], np.cumsum(voxels)])
        for d in range(world):
            matrix[r, d] = (first[d + 1] - first[d], prefix[first[d + 1]] - prefix[first[d]])
    owned = matrix[:, :, 0].sum(axis=0)
    if world > 1:
        assert owned.min() > 0.4 * owned.mean(), f"unbalanced ranges: {owned.tolist()}"
    wide = np.array([r & 1 for r in range(world)], np.uint32)
    plans = []
    for me in range(world):
        out = np.zeros(5 * world + 1, np.uint64)
        assert plan.plancheck_slices(np.ascontiguousarray(matrix).ctypes.data, wide.ctypes.data, world, me, out.ctypes.data) == 0
        plans.append((out[:5 * world].reshape(world, 5).astype(np.int64), int(out[5 * world])))
    longest = {p[1] for p in plans}
    assert len(longest) == 1, "the ranks disagree on the longest send buffer"
    for a in range(world):
        pa = plans[a][0]
        # a's send buffer: slices back to back, in rank order, each the documented size
        assert pa[0, 0] == 0 and np.all(pa[1:, 0] == np.cumsum(pa[:-1, 1]))
        assert pa[:, 1].sum() <= plans[a][1]
        assert np.all(pa[1:, 2] == np.cumsum(pa[:-1, 3]))
        for b in range(world):
            assert pa[b, 1] == plan.plancheck_slice_bytes(int(matrix[a, b, 0]), int(matrix[a, b, 1]), 2 if a & 1 else 1)
            pb = plans[b][0]
            assert pa[b, 1] == pb[a, 3], f"rank {a} sends rank {b} {pa[b, 1]} bytes, rank {b} expects {pb[a, 3]}"
            assert pb[a, 4] == pa[b, 0], f"rank {b} looks for rank {a}'s slice at {pb[a, 4]}, it lies at {pa[b, 0]}"
    # an implausible matrix is refused, not planned
    bad = matrix.copy(); bad[0, 0, 1] = bad[0, 0, 0] * 64 + 1
    out = np.zeros(5 * world + 1, np.uint64)
    assert plan.plancheck_slices(np.ascontiguousarray(bad).ctypes.data, wide.ctypes.data, world, 0, out.ctypes.data) == 1


@pytest.mark.parametrize("world,weights", [(1, 0), (2, 1), (4, 0), (15, 1)])
def test_gather_plan_to_rank_zero(plan, world, weights):
    rng = np.random.default_rng(10 * world + weights)
    sizes = np.zeros((world, 2), np.uint64)
    sizes[:, 0] = rng.integers(0, 50_000, size=world)
    sizes[:, 1] = (sizes[:, 0] * rng.uniform(0, 8, size=world)).astype(np.uint64)
    plans = []
    for me in range(world):
        out = np.zeros(4 * world + 2, np.uint64)
        assert plan.plancheck_gather(sizes.ctypes.data, world, me, weights, out.ctypes.data) == 0
        plans.append(out)
        assert int(out[4 * world + 1]) == int(sizes[:, 0].sum())
        assert int(out[0]) == plan.plancheck_range_bytes(int(sizes[me, 0]), int(sizes[me, 1]), weights)      # everything goes to rank 0
        assert all(int(out[4 * r]) == 0 for r in range(1, world))
    root = plans[0][:4 * world].reshape(world, 4).astype(np.int64)
    assert np.all(root[1:, 1] == np.cumsum(root[:-1, 2])), "rank 0's receive buffer: the ranges back to back, in rank order"
    for r in range(world):
        assert root[r, 2] == int(plans[r][0]) and root[r, 3] == 0
        if r:
            assert np.all(plans[r][:4 * world].reshape(world, 4)[:, 2] == 0), "only rank 0 receives"
    assert len({int(p[4 * world]) for p in plans}) == 1

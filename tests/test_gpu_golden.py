"""The HIP path against the committed golden vectors (tests/golden/*.npz; provenance in make_golden.py)."""
import os

import numpy as np
import pytest

from tests.util import SD_TOL, assert_leaves_match, by_input_index

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.fixture(scope="module")
def chad():
    import chad_tsdf_amd

    chad_tsdf_amd._lib.load()
    return chad_tsdf_amd


def test_g1_morton_codes(chad):
    g = gold("g1_morton")
    assert np.array_equal(chad.morton_codes(g["points"], float(g["res"])), g["codes"])


def test_g3_normals(chad):
    g = gold("g3_normals")
    m = chad.TSDFMap(float(g["res"]), 0.10)
    m.insert(g["points"], g["position"])
    order, codes, nrm = m.last_frame()
    n = len(g["points"])
    assert np.array_equal(by_input_index(order, codes, n), by_input_index(g["order"], g["codes"], n))
    assert np.abs(by_input_index(order, nrm, n) - by_input_index(g["order"], g["normals"], n)).max() <= SD_TOL
    m.close()


def test_g4_ray_voxel_lists(chad):
    g = gold("g4_rays")
    for tag in ("a", "b"):
        res, trunc = (float(x) for x in g[f"params_{tag}"])
        vox, cnt = chad.walk_rays(g["points"], g["position"], res, trunc, 16)
        assert np.array_equal(cnt, g[f"counts_{tag}"]) and np.array_equal(vox, g[f"voxels_{tag}"])


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_g5_leaf_dumps(chad, mode):
    g = gold("g5_leaves")
    for tag in ("cfg1", "cfg2"):
        res, trunc = (float(x) for x in g[f"{tag}_params"])
        m = chad.TSDFMap(res, trunc, update_mode=mode)
        m.insert(g[f"{tag}_points"], g[f"{tag}_position"])
        assert_leaves_match(m.export_leaves(), (g[f"{tag}_mc"], g[f"{tag}_sd"], g[f"{tag}_w"].astype(np.uint32)), trunc)
        m.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_g6_sequence_with_rollover(chad, mode):
    g = gold("g6_sequence")
    m = chad.TSDFMap(0.05, 0.10, update_mode=mode)
    for pts, pos in zip(g["points"], g["positions"]):
        m.insert(pts, pos)
    assert m.closed_submaps == 1
    assert_leaves_match(m.export_leaves(0), (g["closed_mc"], g["closed_sd"], g["closed_w"].astype(np.uint32)), 0.10)
    assert_leaves_match(m.export_leaves(-1), (g["active_mc"], g["active_sd"], g["active_w"].astype(np.uint32)), 0.10)
    m.close()


def test_g8_finalize(chad):
    """The golden submaps enter as bricks holding one observation per leaf (sum = sd, count = 1, so the mean is sd exactly)
    through the merge's entry point; the DAG must come out as frozen."""
    import torch

    g = gold("g8_finalize")
    m = chad.TSDFMap(0.05, 0.10)
    dev = torch.device("cuda", 0)
    for k in range(2):
        mc, sd = g[f"s{k}_mc"], g[f"s{k}_sd"]
        keys, block_of = np.unique(mc & ~np.uint64(63), return_inverse=True)
        shuffle = np.random.default_rng(k).permutation(len(keys))          # archives are unordered: finalize sorts them
        s = np.zeros((len(keys), 64), np.float32)
        c = np.zeros((len(keys), 64), np.int32)
        cell = (mc & np.uint64(63)).astype(np.int64)
        s[block_of, cell] = sd
        c[block_of, cell] = 1
        m.close_submap_with_blocks(torch.from_numpy(keys[shuffle].astype(np.int64)).to(dev), torch.from_numpy(s[shuffle]).to(dev), torch.from_numpy(c[shuffle]).to(dev))
        assert m.dag_roots(k) == tuple(int(x) for x in g[f"s{k}_roots"])
        st = m.dag_stats()
        for l in range(21):
            assert np.array_equal(m.dag_level(l), g[f"s{k}_level{l}"]), (k, l)
            assert (st["uniques"][l], st["dupes"][l]) == tuple(int(x) for x in g[f"s{k}_counts"][l])

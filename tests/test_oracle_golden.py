"""The oracle against the committed golden vectors (tests/golden/*.npz, provenance in make_golden.py) and against
properties of the reference's algorithm that hold independently of any implementation."""
import os

import numpy as np
import pytest

from chad_tsdf_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def test_g1_morton_codes(oracle):
    g = gold("g1_morton")
    assert np.array_equal(oracle.morton_codes(g["points"], float(g["res"])), g["codes"])


def test_morton_layout_and_roundtrip(oracle):
    """libmorton's layout: x -> bit 0, y -> bit 1, z -> bit 2, 21 bits per axis, biased by 2^20 (morton.hpp:21-37)."""
    assert oracle.morton_encode_ivec([[0, 0, 0]])[0] == 0b111 << 60          # bias 2^20 -> bit 20 of each axis -> bits 60..62
    assert oracle.morton_encode_ivec([[1, 0, 0]])[0] == (0b111 << 60) | 0b001
    assert oracle.morton_encode_ivec([[0, 1, 0]])[0] == (0b111 << 60) | 0b010
    assert oracle.morton_encode_ivec([[0, 0, 2]])[0] == (0b111 << 60) | 0b100000
    assert oracle.morton_encode_ivec([[-1, -1, -1]])[0] == (1 << 60) - 1     # 2^20 - 1 on every axis: the 60 low bits
    rng = np.random.default_rng(0)
    v = rng.integers(-(1 << 20), 1 << 20, size=(5000, 3), dtype=np.int64).astype(np.int32)
    mc = oracle.morton_encode_ivec(v)
    assert np.array_equal(oracle.morton_decode_ivec(mc), v)
    assert int(mc.max()) < 1 << 63
    for row in v[:200]:  # BMI2 pdep path == published bit-by-bit interleave
        assert oracle.lib().oracle_morton_encode_portable(int(row[0]), int(row[1]), int(row[2])) == int(oracle.morton_encode_ivec([row])[0])
    # beyond +-2^20 voxels the code aliases silently (morton.hpp:22): same code for v and v + 2^21
    assert oracle.morton_encode_ivec([[5, 6, 7]])[0] == oracle.morton_encode_ivec([[5 + (1 << 21), 6, 7 - (1 << 21)]])[0]


def test_g3_order_and_normals(oracle):
    g = gold("g3_normals")
    order, codes, nrm, starts = oracle.frame_normals(g["points"], g["position"], float(g["res"]))
    assert np.array_equal(order, g["order"]) and np.array_equal(codes, g["codes"])
    assert np.array_equal(starts, g["group_start"])
    assert np.array_equal(nrm, g["normals"])
    # properties: descending codes, ties by input index; unit normals facing the sensor
    assert np.all(codes[1:] <= codes[:-1])
    tie = codes[1:] == codes[:-1]
    assert np.all(order[1:][tie] > order[:-1][tie])
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1).max() < 1e-6
    to_sensor = g["position"][None, :] - g["points"][order]
    assert (np.einsum("ij,ij->i", nrm, to_sensor) >= 0).all()
    # a neighbourhood never crosses a 4x4x4 block, and the last point is alone (normals.hpp:100)
    first = np.flatnonzero(starts)
    assert starts[0] == 1 and starts[-1] == 1
    for a, b in zip(first[:-1], first[1:]):
        assert (codes[a] >> np.uint64(6)) == (codes[b - 1] >> np.uint64(6))


def test_g4_ray_voxel_lists(oracle):
    g = gold("g4_rays")
    for tag in ("a", "b"):
        res, trunc = (float(x) for x in g[f"params_{tag}"])
        vox, cnt = oracle.walk_rays(g["points"], g["position"], res, trunc, 16)
        assert np.array_equal(cnt, g[f"counts_{tag}"]) and np.array_equal(vox, g[f"voxels_{tag}"])
        # properties of the traversal (octree.hpp:86-152)
        recip = np.float32(1.0 / np.float64(np.float32(res)))
        for i in range(0, len(cnt), 7):
            v = vox[i, : cnt[i]].astype(np.int64)
            steps = np.abs(np.diff(v, axis=0))
            assert (steps.sum(axis=1) == 1).all()                      # one axis, one voxel per step
            assert cnt[i] <= 1 + np.abs(v[-1] - v[0]).sum() + 3         # never more than the bounding-box walk
            p, o = g["points"][i], g["position"]
            d = (p - o) / np.float32(np.sqrt(np.float32(((p - o) ** 2).sum())))
            start = np.floor((p - d * np.float32(trunc)) * recip)
            assert np.abs(v[0] - start).max() <= 1                       # starts at the voxel of p - trunc * dir


def test_g5_leaf_dumps(oracle):
    g = gold("g5_leaves")
    for tag in ("cfg1", "cfg2"):
        res, trunc = (float(x) for x in g[f"{tag}_params"])
        m = oracle.OracleMap(res, trunc)
        m.insert(g[f"{tag}_points"], g[f"{tag}_position"])
        mc, sd, w = m.export_leaves()
        assert np.array_equal(mc, g[f"{tag}_mc"]) and np.array_equal(w, g[f"{tag}_w"]) and np.array_equal(sd, g[f"{tag}_sd"])
        assert np.all(np.abs(sd) <= np.float32(trunc) * (1 + 1e-6))      # clamped observations -> clamped mean
        assert int(w.sum()) == int(m.stats()["updates"])


def test_g6_sequence_with_rollover(oracle):
    g = gold("g6_sequence")
    m = oracle.OracleMap(0.05, 0.10)
    for pts, pos in zip(g["points"], g["positions"]):
        m.insert(pts, pos)
    assert m.closed_submaps == 1
    for sub, tag in ((0, "closed"), (-1, "active")):
        mc, sd, w = m.export_leaves(sub)
        assert np.array_equal(mc, g[f"{tag}_mc"]) and np.array_equal(w, g[f"{tag}_w"]) and np.array_equal(sd, g[f"{tag}_sd"])


def test_rollover_rule_is_strictly_greater_than_5m(oracle):
    """tsdf.cpp:52: glm::distance(position, first) > 5.0f -- exactly 5.0 m does not roll over."""
    pts = synth.cfg2_frame(500, seed=3)[0]
    m = oracle.OracleMap(0.05, 0.10)
    m.insert(pts, [0, 0, 0])
    m.insert(pts, [5.0, 0, 0])
    assert m.closed_submaps == 0
    m.insert(pts, [3.0, 4.0 + 1e-5, 0])  # 5.000008 m
    assert m.closed_submaps == 1
    # after a roll-over the comparison is against the pose that triggered it (tsdf.cpp:55-56)
    m.insert(pts, [0, 0, 0])             # 5.000008 m back
    assert m.closed_submaps == 2


def test_running_mean_equals_mean_of_observations(oracle):
    """octree.hpp:161-163 with unit weights: inserting the same scan k times leaves sd unchanged and multiplies w by k."""
    pts, pos = synth.cfg2_frame(4000, seed=77)
    a = oracle.OracleMap(0.05, 0.10)
    a.insert(pts, pos)
    b = oracle.OracleMap(0.05, 0.10)
    for _ in range(3):
        b.insert(pts, pos)
    (mc1, sd1, w1), (mc3, sd3, w3) = a.export_leaves(), b.export_leaves()
    assert np.array_equal(mc1, mc3) and np.array_equal(3 * w1, w3)
    assert np.abs(sd1 - sd3).max() <= 1e-6


def test_empty_and_degenerate_frames(oracle):
    m = oracle.OracleMap(0.05, 0.10)
    m.insert(np.zeros((0, 3), np.float32), [0, 0, 0])
    assert len(m.export_leaves()[0]) == 0
    m.insert(np.array([[np.nan, 0, 0], [0, 0, 0], [np.inf, 1, 1]], np.float32), [0, 0, 0])  # D2: all dropped
    assert len(m.export_leaves()[0]) == 0
    m.insert(np.repeat(np.array([[2.0, 1.0, 0.5]], np.float32), 12, axis=0), [0, 0, 0])     # Q4: NaN normal -> NaN sd, like upstream
    mc, sd, w = m.export_leaves()
    assert len(mc) > 0 and np.isnan(sd).all() and (w == 12).all()


def test_g8_finalize(oracle):
    """Submap::finalize + NodeLevels (submap.hpp:10-106, levels.hpp): frozen level arrays, counters and roots."""
    g = gold("g8_finalize")
    m = oracle.OracleMap(0.05, 0.10)
    for k in range(2):
        m.close_submap_from_leaves(g[f"s{k}_mc"], g[f"s{k}_sd"], g[f"s{k}_w"].astype(np.uint32))
        assert m.dag_roots(k) == tuple(int(x) for x in g[f"s{k}_roots"])
        for l in range(21):
            assert np.array_equal(m.dag_level(l), g[f"s{k}_level{l}"]), (k, l)
            assert m.dag_level_counts(l) == tuple(int(x) for x in g[f"s{k}_counts"][l])
    # structure of the arrays (levels.hpp:56-86): word 0 reserved; a node is its child mask followed by one address per set bit
    top = g["s1_level0"]
    assert top[0] == 0 and bin(int(top[1])).count("1") + 2 <= len(top)
    assert g["s1_level20"][0] == 0 and (g["s1_level20"][1:] == np.uint64(0xFFFFFFFFFFFFFFFF)).sum() == 1   # Q6: one all-ones weight cluster


def test_running_mean_drift_of_the_reference_arithmetic(oracle):
    """The reference's leaf update is a running mean in f32 (octree.hpp:161-163).  Its rounding error against the exact
    mean of the same observations grows with the observation count: negligible at the weights of a single frame, beyond
    the 1e-4 parity bound for voxels observed tens of thousands of times.  (The device keeps sum and count: DESIGN.md.)"""
    res, trunc = 0.2, 0.8
    o = oracle.OracleMap(res, trunc)
    for k in range(40):
        o.insert(synth.box_scan(20_000, (-3, -2, -1), (3, 2, 1), (0.1 + 0.01 * k, 0.0, 0.0), 900 + k), (0.1 + 0.01 * k, 0.0, 0.0))
    mc, sd, w = o.export_leaves()
    drift = np.abs(sd - o.exact_means()) / np.maximum(np.abs(o.exact_means()), trunc)
    assert w.max() > 15_000
    assert drift[w <= 1000].max() < 2e-5
    assert drift[w <= 4096].max() < 5e-5                 # tests/util.py DRIFT_FREE_WEIGHT
    assert drift.max() > 1e-4                            # ... and past it the reference itself is off by more than the bound


def test_chad_grid_is_the_set_of_complete_cells(oracle):
    """The oracle's restatement of ChadGrid (lvr2.cpp:15-133) against a brute-force numpy statement of the same definition: a cell
    is kept iff all eight voxels c + {0,1}^3 hold a leaf; its corner values are the de-quantised bytes in the vertex order of
    lvr2.cpp:92-102."""
    from chad_tsdf_amd.tsdf import decode_morton

    res, trunc = 0.05, 0.10
    pts, pos = synth.cfg2_frame(40_000)
    o = oracle.OracleMap(res, trunc)
    o.insert(pts, pos)
    o.close_submap()
    mc, sd, _ = o.export_leaves(0)
    codes, corners, query_points = o.chad_grid(0)
    assert query_points == len(mc)
    # de-quantised byte per leaf (cluster.hpp:13-26 then :33-52)
    q = (np.clip(sd * (np.float32(1.0) / np.float32(trunc)), -1, 1).astype(np.float32) * np.float32(127) + np.float32(127)).astype(np.uint8)
    back = (q.astype(np.float32) - np.float32(127)) * np.float32(1.0 / 127.0) * np.float32(trunc)
    value = dict(zip(map(tuple, decode_morton(mc).tolist()), back.tolist()))
    order = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    want = {}
    for (x, y, z) in value:
        c = [value.get((x + dx, y + dy, z + dz)) for dx, dy, dz in order]
        if all(v is not None for v in c):
            want[(x, y, z)] = c
    got = dict(zip(map(tuple, decode_morton(codes).tolist()), corners.tolist()))
    assert got.keys() == want.keys() and len(got) > 5_000
    assert all(np.array_equal(np.float32(got[k]), np.float32(want[k])) for k in want)

"""Parity of the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Bar: Morton codes, voxel lists, leaf sets and weights bit-exact; normals and signed
distances within 1e-4 * max(|sd_ref|, sdf_trunc).
"""
import numpy as np
import pytest

from chad_tsdf_amd import synth
from tests.util import SD_TOL, assert_leaves_match, by_input_index

pytestmark = pytest.mark.gpu

MODES = [0, 1, 2]  # CHAD_UPDATE_AUTO, _SCATTER, _TILED


@pytest.fixture(scope="module")
def chad():
    import chad_tsdf_amd

    chad_tsdf_amd._lib.load()  # raises if the in-tree HIP library is missing: no fallback
    return chad_tsdf_amd


def _rand_points(seed, n, scale):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, 3)) * scale).astype(np.float32)


# ---- a3/a4: Morton encode ---------------------------------------------------------------------------------------
def test_morton_codes_bit_exact(chad, oracle):
    pts = np.concatenate([
        _rand_points(1, 4000, 30.0),
        np.array([[0, 0, 0], [-0.0, 0.05, -0.05], [0.049999997, -1e-9, 1e-9], [52428.75, -52428.8, 52428.7],
                  [-52428.8, 52428.75, 0.0], [1e-30, -1e-30, 5.0]], dtype=np.float32),
        (np.arange(-300, 300, dtype=np.float32)[:, None] * np.float32(0.05)).repeat(3, axis=1),  # exact lattice points
    ])
    for res in (0.05, 0.02, 0.1, 0.0371):
        assert np.array_equal(chad.morton_codes(pts, res), oracle.morton_codes(pts, res))


# ---- a5: the sort ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,bits", [(1, 8), (63, 16), (2048, 24), (2049, 40), (100_003, 64), (1_000_000, 22), (3_000_000, 33)])
def test_radix_sort_is_a_stable_sort(chad, n, bits):
    """Replaces std::sort of morton.hpp:85-89 (stable here, so equal codes keep input order: deviation D3).
    3 M keys exceed the resident-grid limit and exercise the ticketed tile assignment."""
    rng = np.random.default_rng(n)
    mask = np.uint64((1 << bits) - 1) if bits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    keys &= mask
    if n > 1000:
        keys[rng.integers(0, n, size=n // 3)] = keys[rng.integers(0, n, size=n // 3)]  # plenty of duplicates
    out, order = chad.sort_pairs(keys, bits)
    want = np.argsort(keys, kind="stable")
    assert np.array_equal(order, want.astype(np.uint32))
    assert np.array_equal(out, keys[want])


# ---- a8: voxel traversal ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("res,trunc", [(0.05, 0.10), (0.02, 0.06), (0.10, 0.20), (0.05, 0.5), (0.07, 0.01)])
def test_ray_walk_voxel_lists_bit_exact(chad, oracle, res, trunc):
    pos = np.array([0.3, -0.2, 0.1], dtype=np.float32)
    axis = np.array([[5, -0.2, 0.1], [0.3, 4, 0.1], [0.3, -0.2, -3], [-5, -0.2, 0.1], [2.3, 1.8, 0.1], [1.3, 0.8, 1.1], [0.3, -0.2, 0.1 + 1e-3]], dtype=np.float32)
    lattice = (np.round(_rand_points(3, 500, 3.0) / res) * res).astype(np.float32)  # points on voxel boundaries
    pts = np.concatenate([_rand_points(2, 3000, 4.0), axis, lattice])
    stride = 96
    gv, gc = chad.walk_rays(pts, pos, res, trunc, stride)
    wv, wc = oracle.walk_rays(pts, pos, res, trunc, stride)
    assert np.array_equal(gc, wc)
    assert gc.max() <= stride
    assert np.array_equal(gv, wv)


# ---- a5/a6/a7: processing order and normals ---------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cfg1", "cfg2_50k", "cfg2_300k", "cfg5_20k"])
def test_normals_match(chad, oracle, name):
    (pts, pos), res, trunc = {
        "cfg1": (synth.cfg1_frame(), 0.10, 0.20),
        "cfg2_50k": (synth.cfg2_frame(50_000), 0.05, 0.10),
        "cfg2_300k": (synth.cfg2_frame(300_000, seed=7), 0.05, 0.10),
        "cfg5_20k": (synth.cfg5_frame(0, 20_000), 0.02, 0.06),
    }[name]
    m = chad.TSDFMap(res, trunc)
    m.insert(pts, pos)
    order, codes, nrm = m.last_frame()
    o_order, o_codes, o_nrm, _ = oracle.frame_normals(pts, pos, res)
    n = len(pts)
    assert sorted(order.tolist()) == sorted(o_order.tolist()) == list(range(n))       # a permutation of the input
    assert np.array_equal(by_input_index(order, codes, n), by_input_index(o_order, o_codes, n))  # codes bit-exact
    # processing order: points grouped by 4x4x4 block, inside a block descending code, ties by ascending input index
    blocks = codes >> np.uint64(6)
    change = np.flatnonzero(blocks[1:] != blocks[:-1]) + 1
    assert len(np.unique(blocks)) == len(change) + 1, "a block is split into several runs"
    same = blocks[1:] == blocks[:-1]
    assert np.all(codes[1:][same] <= codes[:-1][same])
    tie = same & (codes[1:] == codes[:-1])
    assert np.all(order[1:][tie] > order[:-1][tie])
    g, w = by_input_index(order, nrm, n), by_input_index(o_order, o_nrm, n)
    assert np.abs(g - w).max() <= SD_TOL
    m.close()


# ---- a1..a10 end to end: full-frame leaf dumps --------------------------------------------------------------------
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["cfg1", "cfg2_100k", "cfg2_1M", "cfg5_50k", "sphere", "wide_trunc"])
def test_single_frame_leaves(chad, oracle, name, mode):
    rng = np.random.default_rng(5)
    d = rng.standard_normal((60_000, 3))
    sphere = (5.0 * d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)  # the reference demo's shape, main.cpp:9-29
    (pts, pos), res, trunc = {
        "cfg1": (synth.cfg1_frame(), 0.10, 0.20),
        "cfg2_100k": (synth.cfg2_frame(100_000), 0.05, 0.10),
        "cfg2_1M": (synth.cfg2_frame(), 0.05, 0.10),
        "cfg5_50k": (synth.cfg5_frame(3, 50_000), 0.02, 0.06),
        "sphere": ((sphere, np.zeros(3, np.float32)), 0.05, 0.10),
        "wide_trunc": (synth.cfg2_frame(30_000, seed=11), 0.05, 0.40),  # reach of 8 voxels: beyond the LDS tile's halo
    }[name]
    m = chad.TSDFMap(res, trunc, update_mode=mode)
    o = oracle.OracleMap(res, trunc)
    m.insert(pts, pos)
    o.insert(pts, pos)
    m.sync()
    assert_leaves_match(m.export_leaves(), o.export_leaves(), trunc)
    st, ost = m.stats(), o.stats()
    assert st["updates"] == int(ost["updates"])
    assert st["blocks"] == int(ost["blocks"])      # live 4x4x4 blocks == the reference's _node_lookup entries
    assert st["points_used"] == len(pts) and st["overflow"] == 0
    m.close()


def test_full_cfg2_matches_survey_summary(chad):
    """The GPU path reproduces the scalar summary SURVEY.md section 8(c) G7 records from the reference itself."""
    pts, pos = synth.cfg2_frame()
    m = chad.TSDFMap(0.05, 0.10)
    m.insert(pts, pos)
    mc, sd, w = m.export_leaves()
    assert len(mc) == 335_271 and int(w.sum()) == 6_720_936 and m.stats()["blocks"] == 9_409
    m.close()


@pytest.mark.parametrize("mode", MODES)
def test_stream_with_rollover(chad, oracle, mode):
    """cfg3-shaped stream, 20 k points per frame, poses 0.25 m apart: roll-over at frame 21 (5.25 m > 5.0 m)."""
    m = chad.TSDFMap(0.05, 0.10, update_mode=mode)
    o = oracle.OracleMap(0.05, 0.10)
    for k in range(26):
        pts, _ = synth.cfg3_frame(5 * k, 20_000)
        pos = np.array([-12.0 + 0.25 * k, 0.1, 0.0], dtype=np.float32)
        m.insert(pts, pos)
        o.insert(pts, pos)
    m.sync()
    assert m.closed_submaps == o.closed_submaps == 1
    assert_leaves_match(m.export_leaves(0), o.export_leaves(0), 0.10)   # the closed submap (21 frames accumulated)
    assert_leaves_match(m.export_leaves(-1), o.export_leaves(-1), 0.10)  # the active one (5 frames)
    m.close()


def test_device_resident_and_host_inserts_agree(chad):
    import torch

    pts, pos = synth.cfg2_frame(80_000, seed=3)
    a = chad.TSDFMap(0.05, 0.10)
    b = chad.TSDFMap(0.05, 0.10)
    a.insert(pts, pos)
    b.insert(torch.from_numpy(pts).cuda(), pos)
    ga, gb = a.export_leaves(), b.export_leaves()
    assert np.array_equal(ga[0], gb[0]) and np.array_equal(ga[2], gb[2])
    assert np.array_equal(ga[1], gb[1])  # host points and device points: the same map, bit for bit (integer sums)
    a.close(); b.close()


# ---- edge cases ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", MODES)
def test_edge_cases(chad, oracle, mode):
    pos = np.array([0.0, 0.0, 0.0], dtype=np.float32)
    good = synth.cfg2_frame(5_000, seed=9, sensor=(0.0, 0.0, 0.0))[0]
    cases = {
        "empty": np.zeros((0, 3), np.float32),
        "single": np.array([[1.0, 2.0, 0.5]], np.float32),
        "two_same_voxel": np.array([[1.01, 2.01, 0.51], [1.02, 2.02, 0.52]], np.float32),
        "nan_inf_and_sensor": np.concatenate([good[:100], np.array([[np.nan, 0, 0], [np.inf, 1, 1], [0, 0, 0], [1, -np.inf, 2]], np.float32), good[100:200]]),
        "nine_in_one_voxel": (np.array([[2.011, 1.012, 0.513]], np.float32) + np.linspace(0, 0.02, 9, dtype=np.float32)[:, None] * np.array([[1, 0.3, -0.2]], np.float32)),
        "coincident_points": np.repeat(np.array([[2.0, 1.0, 0.5]], np.float32), 12, axis=0),  # Q4: degenerate fit -> NaN in the reference too
        "negative_octant": -np.abs(good[:2000]) - np.float32(0.013),
        "last_point_shares_voxel": np.concatenate([good[:3000], np.repeat(good[:3000].min(axis=0, keepdims=True) - np.float32(0.3), 10, axis=0)
                                                     + np.linspace(0, 0.01, 10, dtype=np.float32)[:, None]]),
    }
    for name, pts in cases.items():
        m = chad.TSDFMap(0.05, 0.10, update_mode=mode)
        o = oracle.OracleMap(0.05, 0.10)
        m.insert(pts, pos)
        o.insert(pts, pos)
        m.sync()
        try:
            assert_leaves_match(m.export_leaves(), o.export_leaves(), 0.10)
        except AssertionError as e:
            raise AssertionError(f"{name}: {e}") from e
        m.close()


def test_last_point_rule(chad, oracle):
    """Q3 (normals.hpp:100): the last point in the reference's global order is never absorbed into a neighbourhood."""
    base = synth.cfg2_frame(4_000, seed=21)[0]
    corner = base.min(axis=0) - np.float32(1.0)  # the smallest code of the frame: a dense voxel of 12 points
    dense = corner + np.linspace(0, 0.03, 12, dtype=np.float32)[:, None] * np.array([[1.0, 0.5, 0.25]], np.float32)
    pts = np.concatenate([base, dense.astype(np.float32)])
    pos = np.array([0.3, -0.2, 0.1], np.float32)
    m = chad.TSDFMap(0.05, 0.10)
    m.insert(pts, pos)
    order, codes, nrm = m.last_frame()
    o_order, o_codes, o_nrm, _ = oracle.frame_normals(pts, pos, 0.05)
    n = len(pts)
    g, w = by_input_index(order, nrm, n), by_input_index(o_order, o_nrm, n)
    assert np.abs(g - w).max() <= SD_TOL
    last = int(o_order[-1])
    to_sensor = (pos - pts[last]) / np.linalg.norm(pos - pts[last])
    assert np.abs(g[last] - to_sensor).max() < 1e-6       # fallback normal for the last point ...
    assert np.abs(g[last] - g[int(o_order[-2])]).max() > 1e-4  # ... unlike the neighbourhood it was cut from
    m.close()


def test_capacity_overflow_is_reported(chad):
    pts, pos = synth.cfg2_frame(50_000)
    # one frame touches ~9 k blocks; max_blocks = 1000 gives a table of 2048 slots (load <= 0.5 at the nominal capacity), and
    # growth only happens between frames: the table fills up, later blocks find no slot and the insert reports it
    m = chad.TSDFMap(0.05, 0.10, max_blocks=1000)
    m.insert(pts, pos)
    with pytest.raises(chad._lib.ChadError) as err:
        m.sync()
    assert err.value.status == chad._lib.ERR_CAPACITY
    assert 1000 < m.stats()["blocks"] <= 2048
    m.close()


def test_reset_and_reuse(chad, oracle):
    pts, pos = synth.cfg2_frame(60_000, seed=4)
    other, pos2 = synth.cfg2_frame(60_000, seed=5, sensor=(1.0, 0.5, 0.0))
    m = chad.TSDFMap(0.05, 0.10)
    m.insert(other, pos2)
    m.reset()
    m.insert(pts, pos)
    o = oracle.OracleMap(0.05, 0.10)
    o.insert(pts, pos)
    assert_leaves_match(m.export_leaves(), o.export_leaves(), 0.10)
    assert m.closed_submaps == 0
    m.close()


def test_save_writes_first_submap(chad, tmp_path):
    pts, pos = synth.cfg2_frame(20_000)
    m = chad.TSDFMap(0.05, 0.10)
    m.insert(pts, pos)
    path = tmp_path / "mesh.ply"
    m.save(str(path))
    assert m.closed_submaps == 1 and len(m.export_leaves(-1)[0]) == 0  # tsdf.cpp:78-81 archives the active submap
    head = path.read_bytes()[:400].decode(errors="replace")
    v, tri = m.mesh(0)                                                 # tsdf.cpp:85: the FIRST submap is reconstructed (tests/test_gpu_readback.py)
    assert f"element vertex {len(v)}" in head and f"element face {len(tri)}" in head and len(tri) > 0
    m.close()


def test_block_pool_grows_between_frames(chad, oracle):
    """A stream that outgrows the initial pool: capacity doubles between frames (the reference's storage is unbounded),
    and the result still matches the oracle."""
    m = chad.TSDFMap(0.05, 0.10, max_blocks=20_000)
    o = oracle.OracleMap(0.05, 0.10)
    for k in range(12):  # 8.7 k blocks after the first frame, 10.7 k (> half) after the second, 17.9 k at the end
        pts, _ = synth.cfg3_frame(8 * k, 20_000)
        pos = np.array([-12.0 + 0.3 * k, 0.1, 0.0], dtype=np.float32)
        m.insert(pts, pos)
        o.insert(pts, pos)
    m.sync()
    st = m.stats()
    assert st["overflow"] == 0 and st["block_capacity"] == 40_000 and st["blocks"] == int(o.stats()["blocks"])
    assert_leaves_match(m.export_leaves(), o.export_leaves(), 0.10)
    m.close()


def test_a_roll_over_on_every_frame(chad, oracle):
    """Poses 6 m apart: every insert closes the previous submap (tsdf.cpp:52-57) while the one before may still be on its way
    into the archive (the roll-over swaps tables and archives behind the frames).  Each closed submap holds exactly one
    frame and must equal the oracle's."""
    m = chad.TSDFMap(0.05, 0.10)
    o = oracle.OracleMap(0.05, 0.10)
    for k in range(6):
        pts, _ = synth.cfg3_frame(40 * k, 30_000)
        pos = np.array([-12.0 + 6.0 * k, 0.1, 0.0], dtype=np.float32)
        m.insert(pts, pos)
        o.insert(pts, pos)
    m.sync()
    assert m.closed_submaps == o.closed_submaps == 5
    for s in (0, 1, 2, 3, 4, -1):
        assert_leaves_match(m.export_leaves(s), o.export_leaves(s), 0.10)
    m.close()


def test_overflow_of_a_closed_submap_stays_reported(chad):
    """The capacity flag is per table (a table's clear may not wipe what a frame in flight reports about another table), and a table
    goes back to the pool when its submap has been archived: what it reported is latched on the host at that point, so a sync at the
    end of a stream still tells that some submap lost blocks -- until the map is reset."""
    pts, pos = synth.cfg2_frame(50_000)
    m = chad.TSDFMap(0.05, 0.10, max_blocks=1000, fixed_capacity=True)
    m.insert(pts, pos)                                             # ~9 k blocks into 2 048 slots: overflows
    far = pos + np.array([6.0, 0.0, 0.0], np.float32)
    small = pts[:200] + np.array([6.0, 0.0, 0.0], np.float32)
    m.insert(small, far)                                           # more than 5 m on: the overflowed submap is closed, its table cleared and pooled
    m.insert(small, far)
    with pytest.raises(chad._lib.ChadError) as err:
        m.sync()
    assert err.value.status == chad._lib.ERR_CAPACITY and m.closed_submaps == 1
    assert m.stats()["blocks"] < 1000                              # the active submap itself is fine
    m.reset()
    m.insert(small, far)
    m.sync()                                                       # nothing reported any more
    m.close()


def test_fast_reciprocal_equals_the_ieee_quotient_for_every_float(chad):
    """The ray set-up's four reciprocals (octree.hpp:92-93: glm::normalize and 1 / dir) are v_rcp_f32 + one Newton step where that is
    the correctly rounded quotient and hipcc's IEEE sequence elsewhere (device_math.hpp).  The claim is checked here by enumeration:
    all 2^32 bit patterns on the device, the library's own inline functions against `1.0f / x`."""
    import ctypes as C

    from chad_tsdf_amd import _lib

    out = (C.c_uint64 * 4)()
    _lib.check(_lib.load().chad_selftest_reciprocal(out), "selftest")
    in_range, bad_fast, bad_any, bad_outside = list(out)
    assert in_range == 2 * 252 * (1 << 23)      # both signs, biased exponents 1 .. 252, every mantissa
    assert bad_fast == 0, f"{bad_fast} inputs inside [2^-126, 2^126) where rcp + Newton is not the IEEE quotient"
    assert bad_any == 0, f"{bad_any} inputs where the kernels' reciprocal differs from 1.0f / x"
    assert bad_outside > 0                      # outside the range the fast sequence IS wrong somewhere: the range test is needed


def test_wave_scans_and_reductions_on_dpp_equal_plain_loops(chad):
    """The prefix sums, suffix minima and reductions over a wave's 64 lanes that the frame kernels and the sort take from the DPP network
    (row shifts and row broadcasts instead of ds_bpermute round trips) against plain loops over the same values, on the device."""
    import ctypes as C

    from chad_tsdf_amd import _lib

    out = (C.c_uint64 * 7)()
    _lib.check(_lib.load().chad_selftest_wave_ops(out), "selftest")
    *bad, checked = list(out)
    assert checked == 512 * 256 * 64
    assert bad == [0, 0, 0, 0, 0, 0], dict(zip(["inclusive sum", "suffix minimum", "minimum", "maximum", "sum", "64-bit minimum"], bad))

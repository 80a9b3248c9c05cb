"""Submap::finalize on the device + host: leaf clusters quantised on the GPU, folded into the
NodeLevels DAG on the host, against the oracle's restatement of submap.hpp:10-106 / levels.hpp / cluster.hpp.

Bar: cluster words, every level array, the uniques/dupes counters and the root addresses bit-exact when the oracle
finalizes the device's own leaves (same sd, same w).  Against the oracle's own insert the signed distances agree to
1e-4 (tests/test_gpu_parity.py), so a byte may fall on the other side of a quantisation step: there the leaf sets must
be identical and every byte within one step."""
import numpy as np
import pytest

from chad_tsdf_amd import synth
from tests.util import assert_leaves_match, clusters_of_leaves

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def chad():
    import chad_tsdf_amd

    chad_tsdf_amd._lib.load()  # raises if the in-tree HIP library is missing: no fallback
    return chad_tsdf_amd


def corridor(frames, n, step=12):
    """Every `step`-th frame of the cfg3 corridor: 0.6 m between poses, a roll-over every ninth frame."""
    return [synth.cfg3_frame(step * k, n) for k in range(frames)]


def assert_same_dag(m, o):
    for l in range(21):
        assert np.array_equal(m.dag_level(l), o.dag_level(l)), f"level {l}"
    st = m.dag_stats()
    for l in range(21):
        assert (st["uniques"][l], st["dupes"][l]) == o.dag_level_counts(l), f"level {l} uniques/dupes"


@pytest.mark.parametrize("res,trunc,n", [(0.05, 0.10, 200_000), (0.02, 0.06, 60_000), (0.10, 0.20, 30_000)])
def test_leaf_clusters_are_the_quantised_leaves(chad, res, trunc, n):
    m = chad.TSDFMap(res, trunc)
    for seed in (5, 6):
        pts, pos = synth.cfg2_frame(n, seed=seed)
        m.insert(pts, pos)
    mc, sd, w = m.export_leaves()
    keys, tsdf, wts = m.export_clusters()
    wk, wt, ww = clusters_of_leaves(mc, sd, w, trunc, saturate=True)
    assert np.array_equal(keys, wk)
    assert np.array_equal(tsdf, wt)
    assert np.array_equal(wts, ww)
    assert (tsdf == np.uint64(0xFFFFFFFFFFFFFFFF)).any() and (tsdf != np.uint64(0xFFFFFFFFFFFFFFFF)).any()   # empty and filled octets


@pytest.mark.parametrize("saturate", [False, True])
def test_dag_equals_the_oracles_finalize_of_the_same_leaves(chad, oracle, saturate):
    m = chad.TSDFMap(0.05, 0.10, dag_saturate_weights=saturate)
    o = oracle.OracleMap(0.05, 0.10)
    o.set_saturate_weights(saturate)
    for pts, pos in corridor(20, 40_000):
        m.insert(pts, pos)
    m.close_submap()                      # TSDFMap::save's hand-over (tsdf.cpp:78-81)
    closed = m.closed_submaps
    assert closed == 3
    for k in range(closed):
        o.close_submap_from_leaves(*m.export_leaves(k))
    assert_same_dag(m, o)
    for k in range(closed):
        assert m.dag_roots(k) == o.dag_roots(k)
    assert m.dag_stats()["submaps"] == closed
    assert m.dag_stats()["dupes"][20] > 0          # clusters are shared across the submaps of one corridor


def test_dag_against_the_oracles_own_inserts(chad, oracle):
    m = chad.TSDFMap(0.05, 0.10)
    o = oracle.OracleMap(0.05, 0.10)
    for pts, pos in corridor(12, 30_000):
        m.insert(pts, pos)
        o.insert(pts, pos)
    m.close_submap()
    o.close_submap()
    assert m.closed_submaps == o.closed_submaps == 2
    trees_equal = 0
    for k in range(2):
        mc, sd = m.dag_leaves(k)
        omc, osd, ow = o.export_leaves(k)
        assert np.array_equal(mc, omc)                                  # the same leaves hang in the tree
        want = clusters_of_leaves(omc, osd, ow, 0.10)[1]
        got = m.export_clusters(k, weights=False)[1]
        step = (got.view(np.uint8).astype(np.int16) - want.view(np.uint8).astype(np.int16))
        assert np.abs(step).max() <= 1                                   # a byte off by one step at most ...
        # ... and rarely.  (3e-3: these 30 k-point frames take the scatter form of the update, whose staging words carry the signed
        # distance in units of trunc / 2^16 -- a byte's step is trunc / 127, so about one mean in a thousand lands on the other
        # side of a step for that alone; with exact sums throughout it was below 1e-3)
        assert (step != 0).mean() < 3e-3
        trees_equal += int(np.array_equal(got, want))
        assert np.abs(sd - osd).max() <= 0.10 / 127 + 1e-5              # cluster.hpp:13-26 truncates: one step
    if trees_equal == 2:                                                # no byte on a step boundary: then everything matches
        assert_same_dag(m, o)
        assert m.dag_roots(1) == o.dag_roots(1)


def test_finalize_is_lazy_in_order_and_leaves_the_insert_loop_alone(chad, oracle):
    m = chad.TSDFMap(0.05, 0.10)
    o = oracle.OracleMap(0.05, 0.10)
    frames = corridor(22, 20_000)
    for pts, pos in frames[:10]:
        m.insert(pts, pos)
    assert m.closed_submaps == 1
    r0 = m.dag_roots(0)                   # folds submap 0 while the second submap is being filled
    before = [m.dag_level(l) for l in range(21)]
    with pytest.raises(chad._lib.ChadError):
        m.dag_roots(1)                    # not closed yet
    for pts, pos in frames[10:]:
        m.insert(pts, pos)
    assert m.closed_submaps == 2
    m.finalize()
    assert m.dag_roots(0) == r0
    for l in range(21):                   # levels only ever grow at the end (addresses stay valid)
        assert np.array_equal(m.dag_level(l)[:len(before[l])], before[l])
    for k in range(2):
        o.close_submap_from_leaves(*m.export_leaves(k))
    assert_same_dag(m, o)
    # the active submap went on undisturbed: it holds the frames since the second roll-over (poses 0.6 m apart: frames 9, 18)
    ref = oracle.OracleMap(0.05, 0.10)
    for pts, pos in frames[18:]:
        ref.insert(pts, pos)
    assert_leaves_match(m.export_leaves(-1), ref.export_leaves(-1), 0.10)


def test_reset_forgets_the_dag_and_an_empty_submap_has_a_root(chad, oracle):
    m = chad.TSDFMap(0.05, 0.10)
    pts, pos = synth.cfg2_frame(30_000, seed=21)
    m.insert(pts, pos)
    m.close_submap()
    assert len(m.dag_level(20)) > 1
    m.reset()
    assert [len(m.dag_level(l)) for l in range(21)] == [1] * 21
    m.insert(np.zeros((0, 3), np.float32), pos)      # a pose, no points
    m.close_submap()
    o = oracle.OracleMap(0.05, 0.10)
    o.insert(np.zeros((0, 3), np.float32), pos)
    o.close_submap()
    assert m.dag_roots(0) == o.dag_roots(0)
    assert_same_dag(m, o)
    assert m.dag_leaves(0)[0].size == 0


def test_full_size_submap_round_trips_through_the_dag(chad):
    """cfg2 at full size (1 M points, 335 k leaves): every leaf comes back out of the DAG, in Morton order, one
    quantisation step from its mean; an identical submap adds nothing to any level."""
    m = chad.TSDFMap(0.05, 0.10)
    pts, pos = synth.cfg2_frame()
    m.insert(pts, pos)
    m.close_submap()
    mc, sd, w = m.export_leaves(0)
    dmc, dsd = m.dag_leaves(0)
    assert np.array_equal(dmc, mc)
    assert np.abs(dsd - sd).max() <= 0.10 / 127 + 1e-6
    words = [len(m.dag_level(l)) for l in range(21)]
    # the same bricks archived once more (a second insert of the scan gives the same bricks too, now that the map's sums are
    # integers -- tests/test_gpu_properties.py::test_maps_are_bit_reproducible -- but this test is about the DAG alone)
    m.close_submap_with_blocks(*m.export_blocks(0, ordered=False))
    assert [len(m.dag_level(l)) for l in range(21)] == words
    assert m.dag_roots(1) == m.dag_roots(0)
    st = m.dag_stats()
    assert st["uniques"][20] + 1 == words[20] and st["ms_device"] > 0


def test_dag_only_mode_releases_closed_bricks_and_keeps_the_dag(chad):
    """CHAD_FLAG_DAG_ONLY: closed submaps live in the DAG alone (what the reference keeps of them, tsdf.cpp:53-54)."""
    import torch

    a = chad.TSDFMap(0.05, 0.10, dag_only=True)
    b = chad.TSDFMap(0.05, 0.10)
    for pts, pos in corridor(20, 20_000):
        a.insert(pts, pos)
        b.insert(pts, pos)
    a.close_submap()
    b.close_submap()
    assert a.closed_submaps == b.closed_submaps == 3
    for k in range(3):
        with pytest.raises(chad._lib.ChadError):
            a.export_leaves(k)                                     # the bricks are gone ...
        amc, asd = a.dag_leaves(k)                                  # ... the tree is not
        bmc, bsd = b.dag_leaves(k)
        assert np.array_equal(amc, bmc) and np.array_equal(amc, b.export_leaves(k)[0])
        assert np.array_equal(asd, bsd)                            # (the maps are bit-reproducible: integer sums)
        assert a.block_count(k) == b.block_count(k)
    assert a.dag_stats()["submaps"] == 3
    # the map goes on working, and what it archives later is released as well
    free_before = torch.cuda.mem_get_info()[0]
    for pts, pos in corridor(30, 20_000)[20:]:
        a.insert(pts, pos)
    a.close_submap()
    a.sync()
    assert a.closed_submaps == 5 and a.dag_stats()["submaps"] == 5
    assert torch.cuda.mem_get_info()[0] >= free_before - (64 << 20)   # no archive copies pile up (they are ~10 MB each here)


@pytest.mark.parametrize("saturate", [False, True])
def test_device_fold_equals_the_host_fold(chad, monkeypatch, saturate):
    """Submap::finalize runs on the device (dag_device.hpp: per level, probe / scan / commit / resolve against tables that live in
    HBM; the upper levels in one launch).  The sequential host fold of dag.hpp (CHAD_DEBUG_DAG_HOST=1), which test_dag_host.py holds
    against the oracle without a GPU, must give the same DAG word for word: all 21 level arrays, the roots of every submap, the
    uniques / dupes counters -- over several submaps that share the DAG, an empty submap first and last, and a stream submap large
    enough for four levels to take the general (many-workgroup) path."""
    def build(host):
        if host:
            monkeypatch.setenv("CHAD_DEBUG_DAG_HOST", "1")
        else:
            monkeypatch.delenv("CHAD_DEBUG_DAG_HOST", raising=False)
        m = chad.TSDFMap(0.05, 0.10, dag_saturate_weights=saturate)
        zero = np.zeros(3, np.float32)
        m.insert(np.zeros((0, 3), np.float32), zero); m.close_submap()            # an empty submap: a childless root pair (submap.hpp:29-44)
        for f in range(2):
            pts, pos = synth.cfg2_frame(250_000, seed=2 + f, sensor=(0.3 + 0.4 * f, -0.2, 0.1))
            m.insert(pts, pos); m.close_submap()
        for k in range(0, 60, 2):
            m.insert(*synth.cfg3_frame(k, 50_000))
        m.close_submap()
        m.insert(np.zeros((0, 3), np.float32), zero); m.close_submap()
        m.finalize()
        out = ([m.dag_level(l) for l in range(21)], [m.dag_roots(s) for s in range(m.closed_submaps)], m.dag_stats())
        m.close()
        return out

    (dl, dr, ds), (hl, hr, hs) = build(False), build(True)
    assert dr == hr and len(dr) == 5
    for l in range(21):
        assert np.array_equal(dl[l], hl[l]), f"level {l}"
    assert list(ds["uniques"]) == list(hs["uniques"]) and list(ds["dupes"]) == list(hs["dupes"])
    assert ds["device_bytes"] > 0


@pytest.mark.parametrize("offset", [(0.0, 0.0, 0.0), (812.0, -3.0, 0.0), (-1640.0, 407.0, 100.0)])
def test_block_key_order_near_and_far_from_the_origin(chad, oracle, offset):
    """The fold wants a submap's blocks in key order, and the sort that puts them there leaves two of its eight passes out when their
    digits are functions of the top digit over the submap's keys (radix_sort.hpp, RadixSortSignals::dep): around the origin the upper
    bits of a coordinate all change with its sign (offset 2^20, morton.hpp:21-37), far from it they are constant -- and where a
    submap lies across a power of two in between (x = 2^14 voxels = 819.2 m here, x = -2^15, y = 2^13) some of them do vary on
    their own and every pass must run.  Both closing routes -- the dense archive, and CHAD_FLAG_DAG_ONLY's cluster words straight
    from the table -- against the oracle's finalize of the same leaves: every level array, the counters, the roots."""
    shift = np.asarray(offset, np.float32)
    frames = [(pts + shift, pos + shift) for pts, pos in corridor(20, 40_000)]
    kept = chad.TSDFMap(0.05, 0.10)
    gone = chad.TSDFMap(0.05, 0.10, dag_only=True)
    o = oracle.OracleMap(0.05, 0.10)
    for pts, pos in frames:
        kept.insert(pts, pos)
        gone.insert(pts, pos)
    kept.close_submap()
    gone.close_submap()
    closed = kept.closed_submaps
    assert closed == gone.closed_submaps == 3
    for k in range(closed):
        o.close_submap_from_leaves(*kept.export_leaves(k))
    assert_same_dag(kept, o)
    assert_same_dag(gone, o)      # (the two maps hold the same integers: test_maps_are_bit_reproducible)
    for k in range(closed):
        assert kept.dag_roots(k) == gone.dag_roots(k) == o.dag_roots(k)
    kept.close(); gone.close()


def test_dag_pool_and_reserved_levels_allocate_nothing_and_change_nothing(chad):
    """chad_options.dag_pool_bytes + chad_map_dag_reserve (the VirtualArray-style reservation, virtual_array.hpp:16-19): a trajectory of
    roll-overs under CHAD_FLAG_DAG_ONLY runs a second time on the same map -- after chad_map_reset, its levels made their final size
    from the start inside the pool -- without a single hipMalloc / hipFree and without a level growing, and builds the same DAG word
    for word as a plain map that grows its levels by hipMalloc."""
    from chad_tsdf_amd import synth

    frames = [(synth.cfg3_frame(5 * k, 20_000)[0], np.array([-12.0 + 0.6 * k, 0.1, 0.0], np.float32)) for k in range(40)]   # a roll-over every nine frames
    plain = chad.TSDFMap(0.05, 0.10, dag_only=True)
    pooled = chad.TSDFMap(0.05, 0.10, dag_only=True, dag_pool_bytes=256 << 20)
    for pts, pos in frames:
        plain.insert(pts, pos); pooled.insert(pts, pos)
    plain.close_submap(); pooled.close_submap()
    plain.finalize(); pooled.finalize()
    want = [plain.dag_level(l) for l in range(21)]
    roots = [plain.dag_roots(s) for s in range(plain.closed_submaps)]
    assert all(np.array_equal(pooled.dag_level(l), want[l]) for l in range(21))
    first = pooled.pipeline_times()
    assert first["dag_pool_bytes"] == 256 << 20 and 0 < first["dag_pool_peak"] <= first["dag_pool_bytes"]
    entries, words = pooled.dag_sizes()
    assert int(words[20]) == len(want[20]) and int(entries[20]) > 100
    pooled.reset()
    # (a fold also wants room for the WORST case of the submap in hand -- every candidate new, nine words per node candidate --, which at
    # this small scale is more than the whole level: the hint leaves that room; at cfg5's scale a sixteenth of the level covers it)
    pooled.dag_reserve(entries * 2, words * 4)
    pooled.sync(); pooled.pipeline_times()
    for pts, pos in frames:
        pooled.insert(pts, pos)
    pooled.close_submap()
    pooled.finalize()
    again = pooled.pipeline_times()
    assert again["mallocs"] == 0 and again["frees"] == 0 and again["level_growths"] == 0, again
    assert again["folds_completed"] == len(roots) and again["step_b_ms"] > 0
    assert all(np.array_equal(pooled.dag_level(l), want[l]) for l in range(21))
    assert [pooled.dag_roots(s) for s in range(len(roots))] == roots
    plain.close(); pooled.close()


def _rollover_frames(n=45):
    from chad_tsdf_amd import synth

    return [(synth.cfg3_frame(5 * k, 20_000)[0], np.array([-12.0 + 0.6 * k, 0.1, 0.0], np.float32)) for k in range(n)]   # a roll-over every nine frames


def test_a_failed_roll_over_step_is_retried_with_the_same_arena(chad, monkeypatch):
    """close_into_dag takes the cluster words' arena by the parity of the roll-overs closed so far.  A step that fails after the
    arena was chosen (an allocation; here injected: CHAD_DEBUG_FAIL_CLOSE) must leave the parity where it was -- the retry, driven by the
    next call, takes the SAME arena and never the one the fold in flight is reading.  The call that hit the failure reports it; the DAG in
    the end is the DAG of an undisturbed run, word for word."""
    frames = _rollover_frames()
    want = chad.TSDFMap(0.05, 0.10, dag_only=True)
    for pts, pos in frames:
        want.insert(pts, pos)
    want.close_submap(); want.finalize()
    monkeypatch.setenv("CHAD_DEBUG_FAIL_CLOSE", "3")   # the third roll-over's step A fails once
    m = chad.TSDFMap(0.05, 0.10, dag_only=True)
    monkeypatch.delenv("CHAD_DEBUG_FAIL_CLOSE")
    errors = []
    for pts, pos in frames:
        try:
            m.insert(pts, pos)
        except chad._lib.ChadError as e:
            errors.append(str(e))   # (the frame of the failing call is not integrated again: see below)
    assert len(errors) == 1 and "CHAD_DEBUG_FAIL_CLOSE" in errors[0]
    m.close_submap(); m.finalize()
    assert m.closed_submaps == want.closed_submaps
    # the failing insert returned before its frame was enqueued: that one frame is missing from its submap, every other submap is untouched
    same = [m.dag_roots(s) == want.dag_roots(s) for s in range(want.closed_submaps)]
    assert sum(same) >= want.closed_submaps - 1 and same[0] and same[1]
    m.close(); want.close()


def test_a_failed_fold_is_abandoned_and_the_map_lives_on(chad, monkeypatch):
    """a fold stage that fails (out of memory for a level that has to grow; here injected: CHAD_DEBUG_FAIL_FOLD) used to leave
    the map re-entering the failed stage on every later call -- it could not even be reset.  Now the fold is abandoned: the error is
    reported once, every DAG reader says the DAG is incomplete, frames and roll-overs go on, and chad_map_reset gives a working map."""
    frames = _rollover_frames()
    monkeypatch.setenv("CHAD_DEBUG_FAIL_FOLD", "4")   # the second roll-over's fold fails in its second stage
    m = chad.TSDFMap(0.05, 0.10, dag_only=True)
    monkeypatch.delenv("CHAD_DEBUG_FAIL_FOLD")
    errors = []
    for pts, pos in frames:
        try:
            m.insert(pts, pos)
        except chad._lib.ChadError as e:
            errors.append(str(e))
    try:
        m.sync()
    except chad._lib.ChadError as e:
        errors.append(str(e))
    assert len(errors) == 1 and "abandoned" in errors[0]
    assert m.stats()["closed_submaps"] == 4 and m.block_count(-1) > 0          # the frames went on, the roll-overs too
    with pytest.raises(chad._lib.ChadError, match="incomplete"):
        m.dag_roots(0)
    with pytest.raises(chad._lib.ChadError, match="incomplete"):
        m.finalize()
    m.reset()                                                                  # ... and the map can be reset: a new DAG
    for pts, pos in frames[:20]:
        m.insert(pts, pos)
    m.close_submap(); m.finalize()
    fresh = chad.TSDFMap(0.05, 0.10, dag_only=True)
    for pts, pos in frames[:20]:
        fresh.insert(pts, pos)
    fresh.close_submap(); fresh.finalize()
    assert m.closed_submaps == fresh.closed_submaps == 3
    assert all(np.array_equal(m.dag_level(l), fresh.dag_level(l)) for l in range(21))
    m.close(); fresh.close()


def test_dag_stats_carries_its_size(chad):
    """chad_dag_stats.struct_size: a caller that does not set it is refused (never a write past its structure), a shorter -- older --
    structure gets the fields it knows."""
    import ctypes as C

    from chad_tsdf_amd import _lib

    m = chad.TSDFMap(0.05, 0.10)
    pts, pos = synth.cfg2_frame(20_000)
    m.insert(pts, pos)
    m.close_submap()
    st = _lib.DagStats()
    assert m._lib.chad_map_dag_stats(m._h, C.byref(st)) == _lib.ERR_INVALID_ARG
    raw = (C.c_uint8 * (C.sizeof(_lib.DagStats) + 16))()
    C.memset(raw, 0xAB, len(raw))
    short = _lib.DagStats.submaps.offset + 8   # struct_size, reserved, submaps
    C.cast(raw, C.POINTER(C.c_uint32))[0] = short
    _lib.check(m._lib.chad_map_dag_stats(m._h, C.cast(raw, C.POINTER(_lib.DagStats))), "dag_stats")
    got = _lib.DagStats.from_buffer_copy(bytes(raw)[:C.sizeof(_lib.DagStats)])
    assert got.struct_size == short and got.submaps == 1
    assert all(b == 0xAB for b in bytes(raw)[short:]), "the library wrote past the caller's size"
    assert m.dag_stats()["submaps"] == 1 and m.dag_stats()["uniques"][20] > 0
    m.close()


"""Size-independent properties of the path, checked at BASELINE.json's full sizes where the oracle would be slow."""
import numpy as np
import pytest

from chad_tsdf_amd import synth
from tests.util import SD_TOL, assert_leaves_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def chad():
    import chad_tsdf_amd

    chad_tsdf_amd._lib.load()
    return chad_tsdf_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle


def rel_err(a, b, trunc):
    return float((np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), trunc)).max())


def test_weights_sum_to_update_count_full_cfg2(chad):
    pts, pos = synth.cfg2_frame()
    m = chad.TSDFMap(0.05, 0.10)
    m.insert(pts, pos)
    mc, sd, w = m.export_leaves()
    st = m.stats()
    assert int(w.sum()) == st["updates"] == 6_720_936
    assert np.all(mc[1:] > mc[:-1]), "leaves are sorted by code and unique"
    assert np.all(np.abs(sd) <= 0.10 * (1 + 1e-6))
    assert st["blocks"] == len(np.unique(mc >> np.uint64(6))) == 9_409
    m.close()


def test_same_scan_twice_doubles_weights_keeps_mean(chad):
    pts, pos = synth.cfg2_frame()
    m = chad.TSDFMap(0.05, 0.10)
    m.insert(pts, pos)
    a = m.export_leaves()
    m.insert(pts, pos)
    b = m.export_leaves()
    assert np.array_equal(a[0], b[0]) and np.array_equal(2 * a[2], b[2])
    assert rel_err(b[1], a[1], 0.10) <= SD_TOL
    m.close()


def test_point_order_does_not_matter(chad):
    """The reference sorts the scan first (morton.hpp:81-102), so a permuted scan gives the same map: leaf set and
    weights exactly; sd up to the tie order of equal codes (f64 summation order in the plane fits)."""
    pts, pos = synth.cfg2_frame(400_000, seed=12)
    perm = np.random.default_rng(1).permutation(len(pts))
    a = chad.TSDFMap(0.05, 0.10)
    b = chad.TSDFMap(0.05, 0.10)
    a.insert(pts, pos)
    b.insert(pts[perm], pos)
    ga, gb = a.export_leaves(), b.export_leaves()
    assert np.array_equal(ga[0], gb[0]) and np.array_equal(ga[2], gb[2])
    # a different tie order changes which point is "last" (normals.hpp:100) and which is first of its neighbourhood
    # (the flip test, normals.hpp:117): a handful of voxels may differ, the rest agrees to rounding
    err = np.abs(ga[1] - gb[1]) / np.maximum(np.abs(gb[1]), 0.10)
    assert (err > SD_TOL).mean() < 1e-3
    a.close(); b.close()


def test_partial_maps_add_up(chad):
    """Linearity (what the multi-GPU merge relies on): blocks(A) + blocks(B) == blocks(A then B)."""
    import torch

    from tests.sharding_standin import merge_blocks

    (pa, pos_a), (pb, pos_b) = synth.cfg2_frame(500_000, seed=21), synth.cfg2_frame(500_000, seed=22, sensor=(0.8, 0.3, -0.2))
    both = chad.TSDFMap(0.05, 0.10)
    both.insert(pa, pos_a)
    both.insert(pb, pos_b)
    ma, mb = chad.TSDFMap(0.05, 0.10), chad.TSDFMap(0.05, 0.10)
    ma.insert(pa, pos_a)
    mb.insert(pb, pos_b)
    ka, sa, ca = ma.export_blocks()
    kb, sb, cb = mb.export_blocks()
    assert torch.all(ka[1:] > ka[:-1]), "export_blocks is sorted by block key"
    union = torch.unique(torch.cat([ka, kb]))
    s = torch.zeros((len(union), 64), device=ka.device)
    c = torch.zeros((len(union), 64), dtype=torch.int32, device=ka.device)
    for k, ss, cc in ((ka, sa, ca), (kb, sb, cb)):
        at = torch.searchsorted(union, k)
        s[at] += ss
        c[at] += cc
    kk, ks, kc = both.export_blocks()
    assert torch.equal(kk, union) and torch.equal(kc, c)
    assert float(((ks - s).abs() / (kc.clamp(min=1) * 0.10)).max()) <= SD_TOL
    # and the archived (merged) submap reads back as leaves
    ma.close_submap_with_blocks(union, s, c)
    mc, sd, w = ma.export_leaves(0)
    bmc, bsd, bw = both.export_leaves(-1)
    assert np.array_equal(mc, bmc) and np.array_equal(w, bw) and rel_err(sd, bsd, 0.10) <= SD_TOL
    assert merge_blocks(ka, sa, ca)[0] is ka  # single process: identity
    for m in (both, ma, mb):
        m.close()


def test_update_kernels_agree_at_full_size(chad):
    pts, pos = synth.cfg2_frame()
    out = []
    for mode in (1, 2):
        m = chad.TSDFMap(0.05, 0.10, update_mode=mode)
        m.insert(pts, pos)
        out.append(m.export_leaves())
        m.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2], out[1][2])
    assert rel_err(out[0][1], out[1][1], 0.10) <= SD_TOL


def test_stream_rollovers_at_expected_frames(chad):
    """cfg3 poses (0.05 m per frame): the 5 m rule closes submaps when frames 101 and 202 arrive."""
    m = chad.TSDFMap(0.05, 0.10, max_blocks=1 << 18)
    total = 0
    closed_at = []
    for k in range(205):
        pts, pos = synth.cfg3_frame(k, 20_000)
        before = m.closed_submaps if k in (100, 101, 201, 202) else None
        m.insert(pts, pos)
        total += m.stats()["updates"]
        if before is not None and m.closed_submaps != before:
            closed_at.append(k)
    assert closed_at == [101, 202]
    w_all = sum(int(m.export_leaves(s)[2].sum()) for s in (0, 1, -1))
    assert w_all == total
    m.close()


def test_sparse_outdoor_frame_and_hash_occupancy(chad):
    """cfg5: 200 k points at 0.02 m -> ~1.6 M voxels in ~330 k blocks, about one point per block: the auto mode takes
    the scatter kernel; the block hash stays at load <= 0.5."""
    pts, pos = synth.cfg5_frame(0)
    m = chad.TSDFMap(0.02, 0.06)
    m.insert(pts, pos)
    mc, sd, w = m.export_leaves()
    st = m.stats()
    assert st["overflow"] == 0 and int(w.sum()) == st["updates"]
    assert st["blocks"] == len(np.unique(mc >> np.uint64(6)))
    assert st["blocks"] / st["hash_capacity"] <= 0.5
    assert 1_500_000 < len(mc) < 1_700_000
    m.close()


def test_cfg3_stream_back_to_back_matches_survey_pin(chad):
    """cfg3 at full size, the way a stream is inserted: 100 frames of 200 k points enqueued back to back (three frames in
    flight, no synchronisation in between), device-resident.  The accumulated submap must be the reference's (SURVEY.md
    8(c) G7, reproduced by the oracle in test_oracle_survey_pins.py): V = 737 935 voxels, sum of weights 135 382 309."""
    import torch

    frames = [synth.cfg3_frame(k) for k in range(100)]
    dev = [torch.from_numpy(p).cuda() for p, _ in frames]
    m = chad.TSDFMap(0.05, 0.10)
    for (_, pos), d in zip(frames, dev):
        m.insert(d, pos)
    m.sync()
    mc, sd, w = m.export_leaves()
    assert m.closed_submaps == 0
    assert len(mc) == 737_935 and int(w.sum(dtype=np.uint64)) == 135_382_309
    assert np.isfinite(sd).all() and np.abs(sd).max() <= 0.10 * (1 + 1e-5)  # a mean of values clamped to +-trunc, summed in f32
    m.close()


def test_sort_survives_a_frame_that_needs_more_passes_than_were_enqueued(chad, oracle):
    """The number of single-pass sort launches follows the previous frames (a hint in host-mapped memory); a frame whose
    extent suddenly needs more key bits is finished by the last enqueued launch, which runs on to the pass limit (radix_sort.hpp, k_radix_onesweep to_limit)."""
    small = synth.cfg2_frame(10_000, seed=3)[0] * np.float32(0.1)       # a 1 m box: few key bits
    pos = np.array([0.03, -0.02, 0.01], dtype=np.float32)
    wide, pos2 = synth.cfg3_frame(0, 60_000)                             # a 30 m corridor: many more
    m = chad.TSDFMap(0.05, 0.10)
    o = oracle.OracleMap(0.05, 0.10)
    for _ in range(3):
        m.insert(small, pos); o.insert(small, pos)
    few = m.stats()["sort_passes"]
    m.insert(wide, pos2 * np.float32(0.0) + pos); o.insert(wide, pos2 * np.float32(0.0) + pos)
    m.sync()
    assert m.stats()["sort_passes"] > few
    got, want = m.export_leaves(), o.export_leaves()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    assert rel_err(got[1], want[1], 0.10) <= SD_TOL
    m.close()


def test_thousands_of_observations_of_one_voxel_in_one_frame(chad, oracle):
    """A scan shrunk to a 0.5 m box puts ~9 k points of one frame into the voxels next to the sensor, each observed at
    +trunc: a tile cell takes 512 observations of the largest fixed-point value per work item (the count field once
    overflowed here).  Weights must be exact; the signed distance of such a voxel is compared loosely, because the
    reference's running mean (octree.hpp:161-163, 9 k sequential f32 updates) drifts by ~2e-4 of trunc itself."""
    pts = synth.cfg2_frame(30_000, seed=3)[0] * np.float32(0.05)
    pos = np.array([0.015, -0.01, 0.005], dtype=np.float32)
    for mode in (1, 2):
        m = chad.TSDFMap(0.05, 0.10, update_mode=mode)
        o = oracle.OracleMap(0.05, 0.10)
        m.insert(pts, pos); o.insert(pts, pos)
        got, want = m.export_leaves(), o.export_leaves()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
        assert int(want[2].max()) > 5_000
        assert rel_err(got[1], want[1], 0.10) <= 1e-3
        few = want[2] < 200   # where the reference's own rounding is small, the usual bar holds
        assert rel_err(got[1][few], want[1][few], 0.10) <= SD_TOL
        m.close()


def test_three_frames_in_flight_all_outgrow_the_enqueued_sort_passes(chad):
    """Three consecutive large frames whose extent needs more sort passes than the previous frames' hint: their grids (489 tiles)
    are too large to be resident beside each other, so each is finished by its catch-all (radix_rest_points, the prologue of the
    k_frame_gather launch that follows the sort: as many of its workgroups as are certainly resident draw tiles by ticket, with a
    barrier among them behind every pass), all three in flight at once.
    Device-resident, back to back, no synchronisation in between."""
    import torch

    small = [torch.from_numpy((synth.cfg2_frame(400_000, seed=31 + k)[0] * np.float32(0.1))).cuda() for k in range(3)]
    wide = [synth.cfg3_frame(k, 1_000_000) for k in range(3)]   # 489 sort tiles each: three such grids exceed what the chip holds at once
    wide_dev = [torch.from_numpy(p).cuda() for p, _ in wide]
    pos = np.array([0.03, -0.02, 0.01], dtype=np.float32)
    m = chad.TSDFMap(0.05, 0.10)
    for d in small:
        m.insert(d, pos)
    m.sync()
    few = m.stats()["sort_passes"]
    for (_, p), d in zip(wide, wide_dev):
        m.insert(d, p)
    m.sync()
    st = m.stats()
    assert st["sort_passes"] > few and st["overflow"] == 0
    # the three wide frames again, one at a time: same voxels, doubled weights
    mc, sd, w = m.export_leaves()
    ref = chad.TSDFMap(0.05, 0.10)
    for d in small:
        ref.insert(d, pos); ref.sync()
    for (_, p), d in zip(wide, wide_dev):
        ref.insert(d, p); ref.sync()
    mc2, sd2, w2 = ref.export_leaves()
    assert np.array_equal(mc, mc2) and np.array_equal(w, w2)
    m.close(); ref.close()


def test_sharded_map_with_one_rank_equals_the_plain_map(chad):
    """ShardedTSDFMap (sharded.py) without a process group is a one-rank world: no communicator is attached and its roll-over is the
    plain map's own; it must leave the same submaps."""
    from chad_tsdf_amd.sharded import ShardedTSDFMap

    plain = chad.TSDFMap(0.05, 0.10)
    shard = ShardedTSDFMap(0.05, 0.10)
    for k in range(26):   # poses 0.25 m apart: roll-over when frame 21 arrives (5.25 m > 5.0 m)
        pts, _ = synth.cfg3_frame(5 * k, 20_000)
        pos = np.array([-12.0 + 0.25 * k, 0.1, 0.0], dtype=np.float32)
        plain.insert(pts, pos)
        shard.insert(pts, pos)
    plain.sync(); shard.local.sync()
    assert plain.closed_submaps == shard.closed_submaps == 1
    for got, want in ((shard.export_leaves(0), plain.export_leaves(0)), (shard.local.export_leaves(-1), plain.export_leaves(-1))):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
        assert rel_err(got[1], want[1], 0.10) <= SD_TOL   # (the merge adds the ranks' f32 partial sums: one rounding per rank)
    plain.close(); shard.local.close()


def test_roll_over_query_is_the_rule_of_the_insert(chad):
    """chad_map_would_roll_over answers tsdf.cpp:50-52 without side effects; the library's own sharding and the protocol restated in
    Python (tests/sharding_standin.py: the roll-over rule in numpy, the merge through export_blocks / close_submap_with_blocks) close
    submaps at the same frames."""
    from chad_tsdf_amd.sharded import ShardedTSDFMap
    from tests.sharding_standin import StandinShardedMap

    m = chad.TSDFMap(0.05, 0.10)
    f32 = lambda *v: np.asarray(v, dtype=np.float32)
    assert not m.would_roll_over(f32(100, 0, 0))                 # no pose yet: the first insert only records it (tsdf.cpp:48)
    m.insert(np.zeros((0, 3), np.float32), f32(1, 2, 3))
    assert not m.would_roll_over(f32(6, 2, 3))                   # exactly 5 m: the rule is "> 5.0f"
    assert m.would_roll_over(f32(6.000001, 2, 3)) and m.would_roll_over(f32(1, 2, -2.01))
    assert m.closed_submaps == 0                                 # asking changed nothing
    m.insert(np.zeros((0, 3), np.float32), f32(6.5, 2, 3))
    assert m.closed_submaps == 1 and not m.would_roll_over(f32(11.4, 2, 3)) and m.would_roll_over(f32(11.6, 2, 3))
    closed = []
    for native in (True, False):
        s = ShardedTSDFMap(0.05, 0.10) if native else StandinShardedMap(chad.TSDFMap(0.05, 0.10))
        for k in range(0, 260, 7):
            pts, pos = synth.cfg3_frame(k, 3000)
            s.insert(pts, pos)
        closed.append((s.closed_submaps, s.local.stats()["blocks"], s.local.export_leaves(-1)[2].sum()))
    assert closed[0] == closed[1] and closed[0][0] == 2


def test_heavily_observed_voxels_follow_the_exact_mean(chad, oracle):
    """Forty frames into a coarse map: voxels with up to 17 000 observations.  There the reference's f32 running mean is
    itself 1.6e-4 off the exact mean of its observations (tests/test_oracle_golden.py::test_running_mean_drift...), so the
    1e-4 bound against it cannot hold for every leaf; what must hold: same leaves, same weights, the device's sum / count
    within 1e-5 of the EXACT mean everywhere, and within 1e-4 of the reference wherever the reference has not drifted."""
    res, trunc = 0.2, 0.8
    m = chad.TSDFMap(res, trunc)
    o = oracle.OracleMap(res, trunc)
    for k in range(40):
        pos = (0.1 + 0.01 * k, 0.0, 0.0)
        pts = synth.box_scan(20_000, (-3, -2, -1), (3, 2, 1), pos, 900 + k)
        m.insert(pts, pos)
        o.insert(pts, pos)
    mc, sd, w = m.export_leaves()
    omc, osd, ow = o.export_leaves()
    exact = o.exact_means()
    assert np.array_equal(mc, omc) and np.array_equal(w, ow) and w.max() > 15_000
    scale = np.maximum(np.abs(exact), trunc)
    assert (np.abs(sd - exact) / scale).max() <= 1e-5
    calm = w <= 4096
    assert (np.abs(sd - osd) / scale)[calm].max() <= SD_TOL
    assert_leaves_match((mc, sd, w), (omc, osd, ow), trunc, allow_drift=True)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_maps_are_bit_reproducible(chad, mode):
    """The map's sums are integers (units of trunc / 2^30), so the order in which the atomics of three frames in flight land
    does not matter: the same frames give the same map, bit for bit -- sums, counts, and the DAG built from them."""
    frames = [synth.cfg2_frame(300_000, seed=70 + k, sensor=(0.3 + 0.02 * k, -0.2, 0.1)) for k in range(4)]
    out = []
    for _ in range(2):
        m = chad.TSDFMap(0.05, 0.10, update_mode=mode)
        for pts, pos in frames:
            m.insert(pts, pos)
        mc, sd, w = m.export_leaves()
        m.close_submap()
        out.append((mc, sd, w, [m.dag_level(l) for l in range(21)]))
    a, b = out
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))        # the means: identical bits
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))


def test_scatter_staging_is_folded_and_read_through(chad, oracle, monkeypatch, capfd):
    """The scatter form stages its observations in packed words (map_kernels.hpp, kPackCountShift): whatever the state of the
    staging -- nothing folded yet, folded in part beside frames in flight, carried across a table growth -- the map read back is
    the oracle's: leaf sets and weights exact, sd within the bar; the device's leaf look-up reads through the staging as well."""
    monkeypatch.setenv("CHAD_DEBUG_FOLD_POINTS", "90000")    # a fold is due every third of these frames
    res, trunc = 0.05, 0.10
    m = chad.TSDFMap(res, trunc, update_mode=1, max_blocks=12_000)   # scatter form throughout; the table grows on the way (7 k blocks after the first frame)
    o = oracle.OracleMap(res, trunc)
    for k in range(9):
        pts, pos = synth.cfg2_frame(40_000, seed=50 + k, sensor=(0.3 + 0.01 * k, -0.2, 0.1))
        m.insert(pts, pos)
        o.insert(pts, pos)
        if k in (0, 4):                                       # read through the staging words, in the middle of the stream
            assert_leaves_match(m.export_leaves(), o.export_leaves(), trunc)
    m.sync()
    mc, sd, w = m.export_leaves()
    assert_leaves_match((mc, sd, w), o.export_leaves(), trunc)
    st = m.stats()
    assert st["blocks"] == int(o.stats()["blocks"]) and st["overflow"] == 0 and st["block_capacity"] > 12_000
    # leaf queries at the corners' own voxels: bit for bit what the export reports
    pick = np.random.default_rng(3).integers(0, len(mc), 5000)
    centres = ((oracle.morton_decode_ivec(mc[pick]).astype(np.float32) + np.float32(0.5)) * np.float32(res)).astype(np.float32)
    qsd, qw = m.query_sd(centres, trilinear=False)
    assert np.array_equal(qw, w[pick]) and np.array_equal(qsd.view(np.uint32), sd[pick].view(np.uint32))
    m.close()
    err = capfd.readouterr().err
    folds = [int(line.split()[-1]) for line in err.splitlines() if line.startswith("chad: staging folds")]
    assert folds and folds[-1] >= 2, err


def test_scatter_staging_folds_at_the_default_threshold(chad, oracle, monkeypatch, capfd):
    """29 sparse frames (200 k points at 2 cm voxels, automatic mode) from one pose, no roll-over: 5.8 M points pass the default fold
    threshold of a three-slot map (5 505 024 = 2^23 less what the frames in flight around a fold can add) once; the map is the
    oracle's whether a voxel's observations sit in the staging word, in the exact sums or in both."""
    monkeypatch.setenv("CHAD_DEBUG_FOLD_POINTS", "5505024")   # the default, spelt out so that the library reports its folds
    res, trunc = 0.02, 0.06
    c5 = synth.CFG5
    m = chad.TSDFMap(res, trunc, max_blocks=1 << 21)
    o = oracle.OracleMap(res, trunc)
    sensor = np.float32([0.0, 0.0, 0.0])
    for k in range(29):
        pts = synth.box_scan(200_000, np.array(c5["lo"]), np.array(c5["hi"]), sensor, 9000 + k)
        m.insert(pts, sensor)
        o.insert(pts, sensor)
    m.sync()
    assert_leaves_match(m.export_leaves(), o.export_leaves(), trunc)
    st = m.stats()
    assert st["overflow"] == 0 and st["blocks"] == int(o.stats()["blocks"])
    m.close()
    folds = [int(line.split()[-1]) for line in capfd.readouterr().err.splitlines() if line.startswith("chad: staging folds")]
    assert folds and folds[-1] >= 1


def test_two_maps_on_two_threads(chad):
    """A map is one host thread's at a time, but distinct maps are independent (the reference's contract): nothing in the library is
    process-wide any more -- the update kernel's launch shape is the map's (resolution, truncation, device), not a static cache.  Two
    threads, each with a map of its own (different voxel sizes, hence different tile shapes) inserting at the same time, must build
    exactly the maps they build alone."""
    import threading

    frames = [synth.cfg2_frame(150_000, seed=30 + k, sensor=(0.2 * k, 0.0, 0.0)) for k in range(6)]
    params = [(0.05, 0.10), (0.04, 0.12)]

    def build(res, trunc, out, key):
        m = chad.TSDFMap(res, trunc)
        for pts, pos in frames:
            m.insert(pts, pos)
        out[key] = m.export_leaves()
        m.close()

    alone, together = {}, {}
    for i, (res, trunc) in enumerate(params):
        build(res, trunc, alone, i)
    threads = [threading.Thread(target=build, args=(res, trunc, together, i)) for i, (res, trunc) in enumerate(params)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        for got, want in zip(together[i], alone[i]):   # codes, signed distances (bit for bit: the sums are integers), weights
            assert got.dtype == want.dtype and np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"map {i}"


def test_sparse_and_dense_frames_interleaved(chad, oracle):
    """In automatic mode the update kernel decides on the device which form a frame takes: tiles in LDS for dense frames, the scatter
    form (staging words, pack-only tables) for sparse ones.  A stream that flips between the kinds puts the hand-overs to the test:
    staged observations under exact sums and the other way round, a table that stops being pack-only in the middle of a submap."""
    res, trunc = 0.02, 0.06
    rng = np.random.default_rng(5)

    def dense_patch(k):   # 150 k points on a 1.5 m x 1 m patch of a wall 3 m away: ~50 points per 4x4x4 block
        y = rng.uniform(-0.75, 0.75, 150_000) + 0.1 * k
        z = rng.uniform(-0.5, 0.5, 150_000)
        x = np.full_like(y, 3.0) + 0.013 * y
        return np.stack([x, y, z], axis=1).astype(np.float32), np.zeros(3, np.float32)

    def sparse_scan(k):
        pts, _ = synth.cfg5_frame(k, 60_000)
        return pts, np.array([0.5 * k, 0.0, 0.0], np.float32)

    kinds = "ssdsddssd"
    m = chad.TSDFMap(res, trunc, max_blocks=1 << 21)
    o = oracle.OracleMap(res, trunc)
    for k, kind in enumerate(kinds):
        pts, pos = sparse_scan(k % 4) if kind == "s" else dense_patch(k)   # (poses within 5 m of the first: one submap)
        m.insert(pts, pos)
        o.insert(pts, pos)
    assert m.closed_submaps == o.closed_submaps == 0
    assert_leaves_match(m.export_leaves(), o.export_leaves(), trunc)
    assert m.stats()["overflow"] == 0
    m.close()

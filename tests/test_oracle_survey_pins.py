"""The oracle against the only external anchors this path has.

Upstream has no tests or fixtures and cannot be built in this image, so parity is UNPINNED in
the strict sense.  SURVEY.md section 8(c) G7 / section 6 however record scalar summaries of a run of the real
reference on the section 8(d) generators: distinct voxels V, sum of weights, 4x4x4 blocks B (hash entries) and
octree nodes.  Reproducing all of them exactly -- including a 100-frame accumulation and the roll-over frame --
with an independently written generator and oracle is strong evidence that both follow the reference.
"""
import numpy as np

from chad_tsdf_amd import synth


def _run(oracle, frames, res, trunc):
    m = oracle.OracleMap(res, trunc)
    for pts, pos in frames:
        m.insert(pts, pos)
    return m


def test_cfg1_plane_scan(oracle):
    m = _run(oracle, [synth.cfg1_frame()], 0.10, 0.20)
    mc, sd, w = m.export_leaves()
    st = m.stats()
    assert len(mc) == 15_256
    assert int(w.sum()) == 698_873
    assert int(st["blocks"]) == 464     # section 6 table
    assert abs(st["updates"] / 100_000 - 6.99) < 0.005
    assert not np.isnan(sd).any()


def test_cfg2_room_scan(oracle):
    m = _run(oracle, [synth.cfg2_frame()], 0.05, 0.10)
    mc, sd, w = m.export_leaves()
    st = m.stats()
    assert len(mc) == 335_271
    assert int(w.sum()) == 6_720_936
    assert int(st["blocks"]) == 9_409
    assert int(st["nodes"]) == 65_739   # octree nodes incl. the root: the pointer tree itself is restated faithfully
    assert int(st["updates"]) == int(w.sum())


def test_cfg3_stream_100_frames_and_rollover(oracle):
    m = oracle.OracleMap(0.05, 0.10)
    for k in range(100):
        m.insert(*synth.cfg3_frame(k))
    mc, sd, w = m.export_leaves()
    assert len(mc) == 737_935           # SURVEY.md section 8(c) G7 "cfg3 @frame 99"
    assert int(w.sum()) == 135_382_309
    assert m.closed_submaps == 0
    m.insert(*synth.cfg3_frame(100))    # 5.0 m from the first pose: NOT greater than 5.0 (tsdf.cpp:52)
    assert m.closed_submaps == 0
    m.insert(*synth.cfg3_frame(101))    # section 8(d): roll-over at k = 101
    assert m.closed_submaps == 1
    # the frame that triggers the roll-over is inserted into the fresh octree (tsdf.cpp:55-70)
    assert int(m.export_leaves()[2].sum()) == int(m.stats()["updates"])

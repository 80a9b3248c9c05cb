"""The exchange's failure paths without a second process: a two-rank communicator over the callback transport whose peer is SILENT
(every collective answers as if the other rank held nothing: zero blocks, zero bytes -- all sizes agree, so every stage of both
exchanges runs to its end; the part of the key space that rank would own is simply lost), and which can be told to fail its n-th
collective.  Checked: a stage that fails is reported ONCE, by
the call that drove it; the closing state is torn down (the map answers stats / sync, keeps inserting, its submap indices stay aligned
with their poses); the communicator is dead afterwards (every later exchange fails at once instead of waiting for peers that will not
come); a map detached from it goes on as a single-GPU map.  The product's multi-rank results are the business of test_gpu_sharded.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class SilentPeerComm:
    def __init__(self, world=2, fail_at=None):
        from chad_tsdf_amd import _lib

        self._lib = _lib.load()
        self.world, self.calls, self.fail_at = world, 0, fail_at

        def all_gather(_user, send, recv, nbytes):
            self.calls += 1
            if self.fail_at is not None and self.calls == self.fail_at:
                return 1
            C.memset(recv, 0, nbytes * world)
            C.memmove(recv, send, nbytes)
            return 0

        def all_reduce(_user, buf, count, dtype):
            self.calls += 1
            if self.fail_at is not None and self.calls == self.fail_at:
                return 1
            return 0   # (the peer adds nothing)

        self._callbacks = (_lib.ALLGATHER_FN(all_gather), _lib.ALLREDUCE_FN(all_reduce))
        self.handle = C.c_void_p()
        _lib.check(self._lib.chad_comm_create_callbacks(0, world, C.cast(self._callbacks[0], C.c_void_p), C.cast(self._callbacks[1], C.c_void_p), None, C.byref(self.handle)), "SilentPeerComm")

    def close(self):
        if self.handle:
            self._lib.chad_comm_destroy(self.handle)
            self.handle = None


def _frames(n):
    from chad_tsdf_amd import synth

    out = []
    for k in range(n):   # poses 0.5 m apart: a roll-over when frame 11, 22, ... arrives
        pts, _ = synth.cfg3_frame(k, 4000)
        out.append((pts, np.array([-12.0 + 0.5 * k, 0.1, 0.0], dtype=np.float32)))
    return out


@pytest.mark.parametrize("dag_only", [True, False])
def test_silent_peer_runs_every_stage(dag_only):
    import chad_tsdf_amd

    comm = SilentPeerComm()
    m = chad_tsdf_amd.TSDFMap(0.05, 0.10, dag_only=dag_only)
    m.attach_comm(comm)
    for pts, pos in _frames(24):
        m.insert(pts, pos)
    m.sync()
    assert m.closed_submaps == 2
    pt = m.pipeline_times()
    assert pt["exchanges"] == 2 and pt["exchange_bytes"] > 0 and pt["comm_world"] == 2 and pt["comm_transport"] == 2
    assert comm.calls == (10 if dag_only else 8)   # sparse: five collectives per exchange; dense: two all-gathers and two all-reduces
    if dag_only:
        assert pt["folds_completed"] == 2 and m.dag_roots(1)[0] > 0
    m.attach_comm(None)
    m.close(); comm.close()


@pytest.mark.parametrize("dag_only,fail_at", [(True, k) for k in range(1, 6)] + [(False, k) for k in range(1, 5)])
def test_a_failed_collective_is_reported_once_and_the_map_lives_on(dag_only, fail_at):
    import chad_tsdf_amd
    from chad_tsdf_amd._lib import ChadError

    comm = SilentPeerComm(fail_at=fail_at)
    m = chad_tsdf_amd.TSDFMap(0.05, 0.10, dag_only=dag_only)
    m.attach_comm(comm)
    frames = _frames(40)
    errors = []
    for k, (pts, pos) in enumerate(frames[:20]):
        try:
            m.insert(pts, pos)
        except ChadError as e:
            errors.append((k, str(e)))
    try:
        m.sync()
    except ChadError as e:
        errors.append(("sync", str(e)))
    assert len(errors) == 1, errors                      # the roll-over at frame 11: its exchange failed in one stage, reported by one call
    assert "exchange" in errors[0][1] and "abandoned" in errors[0][1]
    st = m.stats()                                       # the map answers
    assert st["closed_submaps"] == 1 and st["overflow"] == 0
    assert len(m.submap_positions(0)) == 11              # the lost submap keeps its place and its poses
    # the communicator is dead: the next roll-over's exchange fails at once (no collective is attempted), again reported once
    calls = comm.calls
    with pytest.raises(ChadError, match="aborted"):
        for pts, pos in frames[20:24]:
            m.insert(pts, pos)
        m.sync()
    assert comm.calls == calls
    # detached, the map goes on alone: frames, roll-overs, a clean sync
    m.attach_comm(None)
    for pts, pos in frames[24:40]:
        m.insert(pts, pos)
    m.sync()
    assert m.closed_submaps == 3
    assert m.block_count(-1) > 0
    m.close(); comm.close()

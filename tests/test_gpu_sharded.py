"""The N > 1 path with the real single-device map: two processes share the one GPU of the test box (RCCL refuses that, so the
library's exchange runs over its callback transport, served by gloo), frames dealt round-robin by the library
(chad_map_attach_comm), one merge at the roll-over.  Every rank must end up with the submaps a single process builds from the
whole stream -- bit for bit: the exchange adds 64-bit integer sums.  The RCCL transport itself is exercised with a one-rank
communicator forced through ncclAllGather / ncclAllReduce."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _stream():
    from chad_tsdf_amd import synth

    out = []
    for k in range(26):   # poses 0.25 m apart: roll-over when frame 21 arrives (5.25 m > 5.0 m)
        pts, _ = synth.cfg3_frame(5 * k, 20_000)
        out.append((pts, np.array([-12.0 + 0.25 * k, 0.1, 0.0], dtype=np.float32)))
    return out


def _rank(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chad_tsdf_amd.sharded import ShardedTSDFMap

    sm = ShardedTSDFMap(0.05, 0.10)
    for k, (pts, pos) in enumerate(_stream()):
        sm.insert(pts if k % world == rank else None, pos)
    sm.merge_and_close()   # the active submap (5 frames) as well
    assert sm.closed_submaps == 2
    for s in (0, 1):
        mc, sd, w = sm.export_leaves(s)
        np.savez(os.path.join(outdir, f"rank{rank}_submap{s}.npz"), mc=mc, sd=sd, w=w)
    # Submap::finalize of the merged submaps: every rank folds the same bricks, so the DAGs must be identical word for word
    levels = {f"level{l}": sm.local.dag_level(l) for l in range(21)}
    np.savez(os.path.join(outdir, f"rank{rank}_dag.npz"), roots=np.array([sm.local.dag_roots(0), sm.local.dag_roots(1)], np.uint32), **levels)
    sm.local.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_build_the_single_process_submaps(tmp_path):
    import torch.multiprocessing as mp

    import chad_tsdf_amd

    world, port = 2, 29533
    mp.spawn(_rank, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    single = chad_tsdf_amd.TSDFMap(0.05, 0.10)
    for pts, pos in _stream():
        single.insert(pts, pos)
    single.close_submap()
    for s in (0, 1):
        want = single.export_leaves(s)
        for rank in range(world):
            got = np.load(tmp_path / f"rank{rank}_submap{s}.npz")
            assert np.array_equal(got["mc"], want[0]) and np.array_equal(got["w"], want[2]), f"rank {rank} submap {s}"
            assert np.array_equal(got["sd"].view(np.uint32), want[1].view(np.uint32)), f"rank {rank} submap {s}: N ranks == 1 GPU, bit for bit"
    # ... and the ranks' DAGs: equal to each other, and to the oracle's finalize of the merged leaves
    from oracle import pyoracle

    dags = [np.load(tmp_path / f"rank{rank}_dag.npz") for rank in range(world)]
    o = pyoracle.OracleMap(0.05, 0.10)
    for s in (0, 1):
        got = np.load(tmp_path / f"rank0_submap{s}.npz")
        o.close_submap_from_leaves(got["mc"], got["sd"], got["w"])
    for l in range(21):
        assert np.array_equal(dags[0][f"level{l}"], dags[1][f"level{l}"]), f"level {l} differs between the ranks"
        assert np.array_equal(dags[0][f"level{l}"], o.dag_level(l)), f"level {l} differs from the oracle"
    assert np.array_equal(dags[0]["roots"], dags[1]["roots"]) and tuple(dags[0]["roots"][1]) == o.dag_roots(1)
    single.close()


def test_one_rank_rccl_communicator_goes_through_rccl(monkeypatch):
    """The RCCL transport on the one GPU there is: a one-rank communicator that is NOT short-cut (CHAD_DEBUG_COMM_NO_SHORTCUT), so
    that the roll-over's exchange really calls ncclCommInitRank / ncclAllGather (u64) / ncclAllReduce (i64 and u32 sums) through
    the dlopen-ed librccl.so.1 -- signatures, enum values, stream ordering.  Result: the plain map's submaps, bit for bit."""
    import ctypes as C

    import chad_tsdf_amd
    from chad_tsdf_amd import _lib

    lib = _lib.load()
    monkeypatch.setenv("CHAD_DEBUG_COMM_NO_SHORTCUT", "1")
    uid = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    _lib.check(lib.chad_comm_unique_id(uid), "unique_id")
    comm = C.c_void_p()
    _lib.check(lib.chad_comm_create_rccl(uid, 0, 1, C.byref(comm)), "create_rccl")
    monkeypatch.delenv("CHAD_DEBUG_COMM_NO_SHORTCUT")

    class Handle:
        handle = comm

    plain = chad_tsdf_amd.TSDFMap(0.05, 0.10)
    shard = chad_tsdf_amd.TSDFMap(0.05, 0.10)
    shard.attach_comm(Handle)
    for pts, pos in _stream():
        plain.insert(pts, pos)
        shard.insert(pts, pos)
    plain.close_submap()
    shard.close_submap()          # with a communicator attached: the exchange
    assert plain.closed_submaps == shard.closed_submaps == 2
    for s in (0, 1):
        got, want = shard.export_leaves(s), plain.export_leaves(s)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
        assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    shard.attach_comm(None)
    shard.close(); plain.close()
    lib.chad_comm_destroy(comm)


def test_one_rank_rccl_sparse_exchange_goes_through_send_and_recv(monkeypatch):
    """The sparse exchange's point-to-point step on the RCCL transport, on the one GPU there is: under CHAD_DEBUG_COMM_NO_SHORTCUT a
    one-rank communicator sends its own slice to itself through ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd (comm.hpp,
    exchange_slices) -- the symbols, the int8 dtype and the grouping of the product transport execute, stage 2's slices and stage 4's
    ranges alike.  The map keeps closed submaps as their DAG only (what makes the exchange the sparse one); rank 0's DAG must be the
    plain map's, word for word."""
    import ctypes as C

    import chad_tsdf_amd
    from chad_tsdf_amd import _lib

    lib = _lib.load()
    monkeypatch.setenv("CHAD_DEBUG_COMM_NO_SHORTCUT", "1")
    uid = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    _lib.check(lib.chad_comm_unique_id(uid), "unique_id")
    comm = C.c_void_p()
    _lib.check(lib.chad_comm_create_rccl(uid, 0, 1, C.byref(comm)), "create_rccl")
    monkeypatch.delenv("CHAD_DEBUG_COMM_NO_SHORTCUT")

    class Handle:
        handle = comm

    plain = chad_tsdf_amd.TSDFMap(0.05, 0.10, dag_only=True)
    shard = chad_tsdf_amd.TSDFMap(0.05, 0.10, dag_only=True)
    shard.attach_comm(Handle)
    for pts, pos in _stream():
        plain.insert(pts, pos)
        shard.insert(pts, pos)
    plain.close_submap()
    shard.close_submap()
    plain.sync(); shard.sync()
    assert plain.closed_submaps == shard.closed_submaps == 2
    times = shard.pipeline_times()
    assert times["comm_transport"] == 1 and times["comm_world"] == 1
    assert times["comm_p2p_calls"] >= 4, times   # two submaps, two point-to-point steps each
    assert [shard.dag_roots(s) for s in (0, 1)] == [plain.dag_roots(s) for s in (0, 1)]
    for level in range(21):
        assert np.array_equal(shard.dag_level(level), plain.dag_level(level)), f"level {level}"
    shard.attach_comm(None)
    shard.close(); plain.close()
    lib.chad_comm_destroy(comm)


def _cfg4_rank(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from chad_tsdf_amd import synth

    n_frames = 500
    mine = {k: synth.cfg3_frame(k)[0] for k in range(rank, n_frames, world)}     # a rank only ever sees the points of its own frames
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chad_tsdf_amd.sharded import ShardedTSDFMap

    dev = {k: torch.from_numpy(p).cuda() for k, p in mine.items()}
    torch.cuda.synchronize()
    sm = ShardedTSDFMap(0.05, 0.10)
    for k in range(n_frames):
        sm.insert(dev.get(k), np.asarray((-12.0 + 0.05 * k, 0.1, 0.0), dtype=np.float32))   # no synchronisation: the exchanges ride beside the frames
    sm.merge_and_close()
    sm.local.sync()
    assert sm.closed_submaps == 5 and sm.local.stats()["overflow"] == 0
    if rank == 0:
        for s in range(5):
            mc, sd, w = sm.export_leaves(s)
            np.savez(os.path.join(outdir, f"cfg4_submap{s}.npz"), mc=mc, sd=sd, w=w)
    else:   # the other rank: its submaps must be the same arrays -- a checksum is enough to carry over
        sums = [[int(a.view(np.uint32).astype(np.uint64).sum()) for a in sm.export_leaves(s)] for s in range(5)]
        np.save(os.path.join(outdir, f"cfg4_rank{rank}_sums.npy"), np.array(sums, dtype=np.uint64))
    sm.local.close()
    dist.destroy_process_group()


def test_cfg4_full_stream_on_two_ranks(tmp_path):
    """BASELINE config 4 at the stream's stated size -- 500 frames x 200 k points dealt round-robin, the block partials merged at
    every submap close (roll-overs at frames 101 / 202 / 303 / 404, plus the open submap at the end) -- with the two ranks this box
    can hold (they share its GPU, so the library's exchange runs over its callback transport; RCCL over xGMI is the 8-GPU node's).
    Every closed submap must equal, bit for bit, the submap one process builds from the whole stream."""
    import torch.multiprocessing as mp

    import chad_tsdf_amd
    from chad_tsdf_amd import synth

    world, port = 2, 29541
    mp.spawn(_cfg4_rank, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import torch

    single = chad_tsdf_amd.TSDFMap(0.05, 0.10)
    for k in range(500):
        pts, pos = synth.cfg3_frame(k)
        single.insert(torch.from_numpy(pts).cuda(), pos)
    single.close_submap()
    assert single.closed_submaps == 5
    other = np.load(tmp_path / "cfg4_rank1_sums.npy")
    for s in range(5):
        want = single.export_leaves(s)
        got = np.load(tmp_path / f"cfg4_submap{s}.npz")
        assert np.array_equal(got["mc"], want[0]) and np.array_equal(got["w"], want[2]), f"submap {s}"
        assert np.array_equal(got["sd"].view(np.uint32), want[1].view(np.uint32)), f"submap {s}: two ranks == one GPU, bit for bit"
        assert [int(a.view(np.uint32).astype(np.uint64).sum()) for a in want] == other[s].tolist(), f"submap {s} on rank 1"
    single.close()


def test_sorted_exports_between_the_stages_of_an_exchange(monkeypatch):
    """The exchange keeps the sorted union of the ranks' block keys from its second stage to its third; exports and finalize sort
    block keys too, with the map's one auxiliary sorter.  An export of the ACTIVE submap (sorted bricks, cluster words) issued right
    after the roll-over -- while the exchange has stages outstanding -- must leave the merged submap intact (it used to read the
    sorter's buffer in the third stage).  One-rank communicator that is not short-cut: the three stages really run."""
    import ctypes as C

    import chad_tsdf_amd
    from chad_tsdf_amd import _lib

    lib = _lib.load()
    monkeypatch.setenv("CHAD_DEBUG_COMM_NO_SHORTCUT", "1")
    uid = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    _lib.check(lib.chad_comm_unique_id(uid), "unique_id")
    comm = C.c_void_p()
    _lib.check(lib.chad_comm_create_rccl(uid, 0, 1, C.byref(comm)), "create_rccl")
    monkeypatch.delenv("CHAD_DEBUG_COMM_NO_SHORTCUT")

    class Handle:
        handle = comm

    plain = chad_tsdf_amd.TSDFMap(0.05, 0.10)
    shard = chad_tsdf_amd.TSDFMap(0.05, 0.10)
    shard.attach_comm(Handle)
    for k, (pts, pos) in enumerate(_stream()):
        plain.insert(pts, pos)
        shard.insert(pts, pos)
        if k == 21:   # the frame that closed submap 0 has just been inserted: the exchange is in its first stages
            keys, tsdf, _ = shard.export_clusters(-1)
            bk, bs, bc = shard.export_blocks(-1, ordered=True)
            assert len(keys) == len(bk) > 0 and np.all(np.diff(bk.cpu().numpy().astype(np.uint64)) > 0)
    plain.close_submap()
    shard.close_submap()
    for s in (0, 1):
        got, want = shard.export_leaves(s), plain.export_leaves(s)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
        assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    shard.attach_comm(None)
    shard.close(); plain.close()
    lib.chad_comm_destroy(comm)


def _stopping_rank(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chad_tsdf_amd.sharded import ShardedTSDFMap

    sm = ShardedTSDFMap(0.05, 0.10)
    for k, (pts, pos) in enumerate(_stream()[:22]):   # frame 21 closes submap 0 -- and is the last thing either rank inserts
        sm.insert(pts if k % world == rank else None, pos)
    if rank == 0:
        sm.local.sync()                               # waits for the exchange: needs rank 1's share of every stage
        mc, sd, w = sm.export_leaves(0)
        np.savez(os.path.join(outdir, "stopping_submap0.npz"), mc=mc, sd=sd, w=w)
        sm.local.close()
    else:
        sm.local.close()                              # no sync, no further insert: destroying the map issues this rank's remaining stages
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_that_stops_right_after_a_roll_over_still_completes_the_exchange(tmp_path):
    """The exchange's later stages are enqueued when the host looks (every insert polls).  A rank whose stream ends with the frame
    that triggered the roll-over never inserts again: whatever it calls next that waits -- here nothing but the map's destruction,
    on the other rank a sync -- must drive its share of the exchange to the end, or the peer waits for ever."""
    import torch.multiprocessing as mp

    import chad_tsdf_amd

    world, port = 2, 29547
    mp.spawn(_stopping_rank, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    single = chad_tsdf_amd.TSDFMap(0.05, 0.10)
    for pts, pos in _stream()[:22]:
        single.insert(pts, pos)
    single.sync()
    want = single.export_leaves(0)
    got = np.load(tmp_path / "stopping_submap0.npz")
    assert np.array_equal(got["mc"], want[0]) and np.array_equal(got["w"], want[2])
    assert np.array_equal(got["sd"].view(np.uint32), want[1].view(np.uint32))
    single.close()


def _cfg5_dag_rank(rank, world, port, outdir, n_frames):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chad_tsdf_amd import synth
    from chad_tsdf_amd.sharded import ShardedTSDFMap

    dev = {k: synth.cfg5_frame_torch(k)[0] for k in range(rank, n_frames, world)}   # a rank only ever sees the points of its own frames
    torch.cuda.synchronize()
    sm = ShardedTSDFMap(0.02, 0.06, max_points=200_000, max_blocks=1 << 22, dag_only=True)
    sm.local.pipeline_times()
    for k in range(n_frames):
        sm.insert(dev.get(k), np.array([0.5 * k, 0.0, 0.0], dtype=np.float32))   # no synchronisation: the exchanges ride beside the frames
    sm.merge_and_close()
    sm.local.sync()
    sm.local.finalize()
    assert sm.local.stats()["overflow"] == 0
    pt = sm.local.pipeline_times()
    np.save(os.path.join(outdir, f"cfg5_rank{rank}_exchange.npy"),
            np.array([pt["exchanges"], pt["exchange_bytes"], pt["exchange_union_blocks"], pt["folds_completed"], pt["comm_world"], pt["comm_transport"]], dtype=np.uint64))
    if rank == 0:
        n_sub = sm.closed_submaps
        levels = {f"level{l}": sm.local.dag_level(l) for l in range(21)}
        np.savez(os.path.join(outdir, "cfg5_rank0_dag.npz"), roots=np.array([sm.local.dag_roots(s) for s in range(n_sub)], np.uint32), **levels)
    else:   # the other ranks hold no DAG: the fold ran once, on rank 0
        from chad_tsdf_amd._lib import ChadError

        with pytest.raises(ChadError, match="rank 0"):
            sm.local.dag_roots(0)
    sm.local.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_cfg5_sharded_dag_only_equals_one_gpu_word_for_word(tmp_path, world):
    """BASELINE config 5's multi-GPU shape on the ranks this box can hold (they share its GPU: callback transport over gloo): 110 frames
    of the outdoor trajectory -- ten roll-overs by the 5 m rule, plus the open submap closed at the end -- dealt round-robin over 2 and
    4 ranks under CHAD_FLAG_DAG_ONLY.  The exchange at a roll-over is the SPARSE one (exchange_kernels.hpp: keys, voxel masks and one
    staging word per touched voxel to the owner of the Morton key range; cluster words of each range to rank 0) and Submap::finalize
    runs ONCE per submap, on rank 0.  Rank 0's DAG must equal the DAG one GPU builds from the whole stream: all 21 level arrays and
    every submap's roots, word for word (submap.hpp:10-106, levels.hpp:141-199).  Bytes a rank hands to the exchange per roll-over are
    printed and bounded: the dense all-reduce of round 3 moved 1.27 GB per rank and roll-over here."""
    import torch.multiprocessing as mp

    import chad_tsdf_amd
    from chad_tsdf_amd import synth

    n_frames = 110
    mp.spawn(_cfg5_dag_rank, args=(world, 29553 + world, str(tmp_path), n_frames), nprocs=world, join=True)
    single = chad_tsdf_amd.TSDFMap(0.02, 0.06, max_points=200_000, max_blocks=1 << 22, dag_only=True)
    for k in range(n_frames):
        pts, pos = synth.cfg5_frame_torch(k)
        single.insert(pts, pos)
    single.close_submap()
    single.finalize()
    n_sub = single.closed_submaps
    assert n_sub == 10
    got = np.load(tmp_path / "cfg5_rank0_dag.npz")
    assert np.array_equal(got["roots"], np.array([single.dag_roots(s) for s in range(n_sub)], np.uint32)), "roots differ from the single-GPU DAG"
    for l in range(21):
        assert np.array_equal(got[f"level{l}"], single.dag_level(l)), f"level {l} differs from the single-GPU DAG"
    single.close()
    per_rollover = []
    for rank in range(world):
        exchanges, sent, union_blocks, folds, comm_world, transport = np.load(tmp_path / f"cfg5_rank{rank}_exchange.npy").tolist()
        assert exchanges == n_sub and comm_world == world and transport == 2
        assert folds == (n_sub if rank == 0 else 0), "Submap::finalize must run once per submap, on rank 0 only"
        per_rollover.append(sent / exchanges / 1e6)
    print(f"world {world}: MB handed to the exchange per roll-over and rank: {[round(b, 1) for b in per_rollover]} (last union: {union_blocks} blocks)")
    assert max(per_rollover) <= 80.0 and sum(per_rollover) / world <= 60.0


def _mixed_stream():
    """_stream() with one frame of 300 k points among the 20 k-point ones: in the scatter form a frame of more than 2^18 points adds to the
    EXACT rows (two atomics per observation), the others to the staging words -- the rank that integrates it sends two words per voxel,
    the other rank one."""
    from chad_tsdf_amd import synth

    out = []
    for k in range(26):
        pts, _ = synth.cfg3_frame(5 * k, 300_000 if k == 3 else 20_000)
        out.append((pts, np.array([-12.0 + 0.25 * k, 0.1, 0.0], dtype=np.float32)))
    return out


def _mixed_rank(rank, world, port, outdir, update_mode):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chad_tsdf_amd.sharded import ShardedTSDFMap

    sm = ShardedTSDFMap(0.05, 0.10, dag_only=True, dag_saturate_weights=True, update_mode=update_mode)
    for k, (pts, pos) in enumerate(_mixed_stream()):
        sm.insert(pts if k % world == rank else None, pos)
    sm.merge_and_close()
    sm.local.sync()
    if rank == 0:
        levels = {f"level{l}": sm.local.dag_level(l) for l in range(21)}
        np.savez(os.path.join(outdir, "mixed_rank0_dag.npz"), roots=np.array([sm.local.dag_roots(s) for s in range(2)], np.uint32), **levels)
    sm.local.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("update_mode", [1, 2])
def test_sparse_exchange_with_exact_rows_and_weight_words(tmp_path, update_mode):
    """The sparse exchange's other payload shapes: two words per voxel (a table whose exact rows are in use: every frame under the tiled
    update, update_mode 2; under the scatter form, update_mode 1, only the rank that integrates the 300 k-point frame -- so the two ranks
    send DIFFERENT widths) and the weight clusters travelling beside the signed-distance clusters (CHAD_FLAG_DAG_SATURATE_WEIGHTS).
    Rank 0's DAG == the single GPU's, word for word."""
    import torch.multiprocessing as mp

    import chad_tsdf_amd

    world = 2
    mp.spawn(_mixed_rank, args=(world, 29571 + update_mode, str(tmp_path), update_mode), nprocs=world, join=True)
    single = chad_tsdf_amd.TSDFMap(0.05, 0.10, dag_only=True, dag_saturate_weights=True, update_mode=update_mode)
    for pts, pos in _mixed_stream():
        single.insert(pts, pos)
    single.close_submap()
    single.finalize()
    assert single.closed_submaps == 2
    got = np.load(tmp_path / "mixed_rank0_dag.npz")
    assert np.array_equal(got["roots"], np.array([single.dag_roots(s) for s in range(2)], np.uint32))
    for l in range(21):
        assert np.array_equal(got[f"level{l}"], single.dag_level(l)), f"level {l} differs from the single-GPU DAG"
    assert len(single.dag_level(20)) > 1000
    single.close()

"""Randomised soak of the SPARSE multi-GPU exchange (not part of the test suite): random resolutions, frame sizes, sensor steps,
update modes, weight modes and 2-4 ranks sharing this box's one GPU (callback transport over gloo) -- some submaps empty, some ranks
without a single block at a roll-over, big frames that take the exact rows on one rank only -- against ONE map fed the whole stream.
Rank 0's DAG must equal the single map's word for word: every level array, every submap's roots.
  python tests/soak/fuzz_sharded_dag.py [seconds] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np


def case(seed):
    rng = np.random.default_rng(seed)
    res = float(np.float32(rng.choice([0.02, 0.05, 0.1])))
    return dict(res=res, trunc=float(np.float32(res * rng.choice([2.0, 3.0]))), frames=int(rng.choice([9, 20, 45])), n=int(rng.choice([0, 60, 3000, 40_000, 200_000])),
                step=float(rng.uniform(0.6, 4.0)), shrink=np.float32(rng.choice([0.1, 1.0, 3.0]) * res / 0.05), offset=np.asarray(rng.choice([0.0, 700.0, -1650.0]) * np.array([1.0, rng.uniform(-1, 1), 0.0]), np.float32),
                saturate=bool(rng.integers(2)), update_mode=int(rng.choice([0, 1, 2])), world=int(rng.choice([2, 3, 4])), big=int(rng.integers(0, 9)))


def frame(c, seed, k):
    from chad_tsdf_amd import synth

    n = 300_000 if (k == c["big"] and c["n"] >= 3000) else c["n"]
    pts, _ = synth.cfg2_frame(max(n, 1), seed=seed * 1000 + k)
    pos = np.array([c["step"] * k, 0.1 * k, 0.0], np.float32)
    return ((pts[:n] * c["shrink"]).astype(np.float32) + pos + c["offset"]).astype(np.float32), pos + c["offset"]


def rank_main(rank, world, port, outdir, seed):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chad_tsdf_amd.sharded import ShardedTSDFMap

    c = case(seed)
    sm = ShardedTSDFMap(c["res"], c["trunc"], dag_only=True, dag_saturate_weights=c["saturate"], update_mode=c["update_mode"])
    for k in range(c["frames"]):
        pts, pos = frame(c, seed, k)
        sm.insert(pts if k % world == rank else None, pos)
    sm.merge_and_close()
    sm.local.sync()
    assert sm.local.stats()["overflow"] == 0
    if rank == 0:
        n_sub = sm.closed_submaps
        np.savez(os.path.join(outdir, f"dag_{seed}.npz"), roots=np.array([sm.local.dag_roots(s) for s in range(n_sub)], np.uint32).reshape(n_sub, 2),
                 **{f"level{l}": sm.local.dag_level(l) for l in range(21)})
    sm.local.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import tempfile

    import torch.multiprocessing as mp

    import chad_tsdf_amd as chad

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t_end = time.time() + budget
    cases = submaps = 0
    with tempfile.TemporaryDirectory() as outdir:
        while time.time() < t_end:
            c = case(seed)
            mp.spawn(rank_main, args=(c["world"], 30000 + seed % 2000, outdir, seed), nprocs=c["world"], join=True)
            single = chad.TSDFMap(c["res"], c["trunc"], dag_only=True, dag_saturate_weights=c["saturate"], update_mode=c["update_mode"])
            for k in range(c["frames"]):
                single.insert(*frame(c, seed, k))
            single.close_submap()
            single.finalize()
            got = np.load(os.path.join(outdir, f"dag_{seed}.npz"))
            n_sub = single.closed_submaps
            assert len(got["roots"]) == n_sub, (seed, c)
            for s in range(n_sub):
                assert tuple(got["roots"][s]) == single.dag_roots(s), f"seed {seed} {c}: roots of submap {s}"
            for l in range(21):
                assert np.array_equal(got[f"level{l}"], single.dag_level(l)), f"seed {seed} {c}: level {l}"
            single.close()
            cases += 1; submaps += n_sub; seed += 1
            if cases % 5 == 0:
                print(f"... {cases} cases, {submaps} submaps, next seed {seed}", flush=True)
    print(f"fuzz ok: {cases} cases, {submaps} submaps, seeds up to {seed - 1}")

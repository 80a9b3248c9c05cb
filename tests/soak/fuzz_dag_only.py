"""Randomised soak of CHAD_FLAG_DAG_ONLY roll-overs (not part of the test suite): random resolutions, scene sizes, frame sizes and
sensor steps -- roll-overs every few frames, some submaps empty, some of a handful of blocks, some large enough for the DAG's levels to
grow -- into two maps, one that keeps its closed submaps as dense archives and folds them when asked, one that keeps the DAG only
(cluster words straight from the table beside the fold of the submap before: chad_hip.hip, close_into_dag).  Both must hold the same
DAG word for word: every level array, the counters, every submap's roots and leaves.
  python tests/soak/fuzz_dag_only.py [seconds] [first seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import chad_tsdf_amd as chad
from chad_tsdf_amd import synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
cases = submaps = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    res = float(np.float32(rng.choice([0.02, 0.05, 0.1])))
    trunc = float(np.float32(res * rng.choice([2.0, 3.0])))
    frames = int(rng.choice([6, 15, 40]))
    n = int(rng.choice([0, 50, 3000, 40_000, 200_000]))
    step = float(rng.uniform(0.6, 4.0))                      # metres between poses: a roll-over every 2 to 9 frames
    shrink = np.float32(rng.choice([0.1, 1.0, 3.0]) * res / 0.05)
    offset = np.asarray(rng.choice([0.0, 700.0, -1650.0]) * np.array([1.0, rng.uniform(-1, 1), 0.0]), np.float32)
    saturate = bool(rng.integers(2))
    kept = chad.TSDFMap(res, trunc, dag_saturate_weights=saturate)
    gone = chad.TSDFMap(res, trunc, dag_saturate_weights=saturate, dag_only=True)
    for k in range(frames):
        pts, _ = synth.cfg2_frame(max(n, 1), seed=seed * 1000 + k)
        pts = (pts[:n] * shrink).astype(np.float32)
        pos = np.array([step * k, 0.1 * k, 0.0], np.float32)
        pts = pts + pos + offset
        for m in (kept, gone):
            m.insert(pts, pos + offset)
        if rng.random() < 0.15:
            gone.sync()                                      # (frees what the DAG's levels have outgrown; finishes folds in flight)
        if rng.random() < 0.1:
            kept.dag_stats()                                 # folds the archives closed so far
    kept.close_submap(); gone.close_submap()
    closed = kept.closed_submaps
    assert closed == gone.closed_submaps, (seed, closed, gone.closed_submaps)
    for l in range(21):
        assert np.array_equal(kept.dag_level(l), gone.dag_level(l)), f"seed {seed}: level {l}"
    sk, sg = kept.dag_stats(), gone.dag_stats()
    assert sk["uniques"] == sg["uniques"] and sk["dupes"] == sg["dupes"], f"seed {seed}: counters"
    for s in range(closed):
        assert kept.dag_roots(s) == gone.dag_roots(s), f"seed {seed}: roots of submap {s}"
    for s in sorted(set([0, closed // 2, closed - 1])):
        a, b = kept.dag_leaves(s), gone.dag_leaves(s)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), f"seed {seed}: leaves of submap {s}"
    assert gone.stats()["overflow"] == 0
    kept.close(); gone.close()
    cases += 1; submaps += closed; seed += 1
    if cases % 10 == 0:
        print(f"... {cases} cases, {submaps} submaps, next seed {seed}", flush=True)
print(f"fuzz ok: {cases} cases, {submaps} submaps, seeds up to {seed - 1}")

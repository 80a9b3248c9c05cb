"""Randomised parity sweep (not part of the test suite: a longer soak for the GPU box).  Random resolutions, truncation
ratios, scene boxes, sensor poses, frame sizes and frame counts; every map is compared with the oracle's (leaf sets and
weights exact, sd within 1e-4 of max(|sd|, trunc)) and its closed submaps' DAG with the oracle's finalize of the same leaves.
  python tests/soak/fuzz_parity.py [seconds] [first seed] [big]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import chad_tsdf_amd as chad
from chad_tsdf_amd import synth
from oracle import pyoracle
from tests.util import assert_leaves_match

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = len(sys.argv) > 3 and sys.argv[3] == "big"   # frames of 1-4 M points (the oracle needs 0.15-0.6 s for each)
t_end = time.time() + budget
cases = 0
last_report = time.time()
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    res = float(np.float32(rng.choice([0.02, 0.035, 0.05, 0.0731, 0.1, 0.2])))
    trunc = float(np.float32(res * rng.choice([1.0, 1.5, 2.0, 3.0, 4.0])))
    half = rng.uniform(1.0, 12.0, size=3) * (res / 0.05)
    lo, hi = -half, half * rng.uniform(0.3, 1.0, size=3)
    frames = int(rng.choice([1, 1, 2, 5, 12]))
    n = int(rng.choice([1, 7, 64, 1000, 20_000, 150_000, 400_000]))
    if big:
        n, frames = int(rng.choice([1_000_000, 2_000_000, 4_000_000])), int(rng.choice([1, 2, 3]))
    step = rng.uniform(0.0, 1.5) * (res / 0.05)
    mode = int(rng.choice([0, 0, 1, 2]))
    m = chad.TSDFMap(res, trunc, update_mode=mode, dag_saturate_weights=bool(seed & 1))
    o = pyoracle.OracleMap(res, trunc)
    o.set_saturate_weights(bool(seed & 1))
    for k in range(frames):
        sensor = np.clip(np.array([lo[0] * 0.5 + step * k, 0.1 * hi[1], 0.0]), lo + 1e-3, hi - 1e-3)
        pts = synth.box_scan(n, lo, hi, sensor, 7000 + 31 * seed + k)
        if rng.random() < 0.2 and n > 10:   # unusable points sprinkled in (D2): NaN, inf, the sensor itself
            bad = rng.integers(0, n, size=max(1, n // 50))
            pts[bad[: len(bad) // 3]] = np.nan
            pts[bad[len(bad) // 3: 2 * len(bad) // 3], 1] = np.inf
            pts[bad[2 * len(bad) // 3:]] = sensor.astype(np.float32)
        pos = sensor.astype(np.float32)
        m.insert(pts, pos)
        o.insert(pts, pos)
    what = f"seed {seed}: res {res:g} trunc {trunc:g} frames {frames} x {n} pts, step {step:.2f} m, mode {mode}"
    try:
        assert m.closed_submaps == o.closed_submaps, "closed submaps"
        for s in list(range(o.closed_submaps)) + [-1]:
            # allow_drift: leaves with more than 4 096 observations are compared with a bound that grows with their weight --
            # the reference's f32 running mean is itself off the exact mean there (tests/util.py)
            assert_leaves_match(m.export_leaves(s), o.export_leaves(s), trunc, allow_drift=True)
        got = m.export_leaves(-1)
        exact = o.exact_means()   # the device's own accuracy: sum / count against the f64 mean of the same observations
        if got[1].size:
            # (the map's sums are exact integers: no allowance for the weight here; what remains is the tiled kernel's incremental
            # distance along a ray, up to 1.4e-5 of a truncation distance of one voxel, per observation)
            e = np.abs(got[1] - exact) / np.maximum(np.abs(exact), trunc)
            assert e.max() <= 3e-5, f"sd vs exact mean: {e.max():.3g} at weight {got[2][e.argmax()]} (reference there: {abs(o.export_leaves(-1)[1][e.argmax()] - exact[e.argmax()]) / max(abs(exact[e.argmax()]), trunc):.3g})"
        m.close_submap()
        ref = pyoracle.OracleMap(res, trunc)
        ref.set_saturate_weights(bool(seed & 1))
        for s in range(m.closed_submaps):
            ref.close_submap_from_leaves(*m.export_leaves(s))
        for l in range(21):
            assert np.array_equal(m.dag_level(l), ref.dag_level(l)), f"DAG level {l}"
    except AssertionError as e:
        print("FAIL", what, "->", e)
        sys.exit(1)
    cases += 1
    seed += 1
    if time.time() - last_report > 60.0:   # (a run that writes nothing for minutes is taken to be hung by the GPU box's watchdog)
        print(f"... {cases} cases so far, next seed {seed}", flush=True)
        last_report = time.time()
    m.close()
print(f"fuzz ok: {cases} cases, seeds up to {seed - 1}")

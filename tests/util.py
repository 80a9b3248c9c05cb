"""Shared helpers of the parity tests."""
import numpy as np

SD_TOL = 1e-4  # north_star: per-voxel SDF within 1e-4 relative, with the truncation distance as the scale floor


DRIFT_FREE_WEIGHT = 4096  # up to this many observations of a voxel the reference's f32 running mean stays within ~3e-5 of the exact mean


def assert_leaves_match(got, want, trunc, tol=SD_TOL, allow_drift=False):
    """Leaf sets (codes) and weights bit-exact; sd within tol * max(|sd_ref|, trunc).

    allow_drift: the reference's leaf is a running mean in f32 (octree.hpp:161-163: sd = (sd * w + d) / (w + 1)), whose
    rounding error grows roughly linearly with the observation count -- 5e-6 at 1 000 observations, 5e-5 at 8 000, 2-3e-4 at
    40-75 000 (tests/test_oracle_golden.py::test_running_mean_drift...).  The device's exact integer sums cannot follow it there -- they
    stay at the exact mean --, so soak tests over long accumulations widen the bound against the reference for such leaves
    in proportion to their weight."""
    gmc, gsd, gw = got
    wmc, wsd, ww = want
    assert gmc.shape == wmc.shape, f"leaf count {gmc.shape[0]} != {wmc.shape[0]}"
    assert np.array_equal(gmc, wmc), "Morton codes of the leaf sets differ"
    assert np.array_equal(gw, ww), f"weights differ at {np.flatnonzero(gw != ww)[:5]}"
    nan_g, nan_w = np.isnan(gsd), np.isnan(wsd)
    assert np.array_equal(nan_g, nan_w), "NaN signed distances (degenerate neighbourhoods) differ"
    ok = ~nan_w
    err = np.abs(gsd[ok].astype(np.float64) - wsd[ok]) / np.maximum(np.abs(wsd[ok]), trunc)
    if allow_drift:
        err = err / np.maximum(1.0, ww[ok] / float(DRIFT_FREE_WEIGHT))
    assert err.size == 0 or err.max() <= tol, f"sd error {err.max():.3g} > {tol}"
    return 0.0 if err.size == 0 else float(err.max())


def by_input_index(order, values, n):
    out = np.zeros((n,) + values.shape[1:], dtype=values.dtype)
    out[order] = values
    return out


def clusters_of_leaves(mc, sd, w, trunc, saturate=False):
    """What k_leaf_clusters delivers: block keys ascending and eight cluster words per block (cluster.hpp:13-31)."""
    recip = np.float32(1.0) / np.float32(trunc)
    q = np.clip(sd.astype(np.float32) * recip, np.float32(-1), np.float32(1)) * np.float32(127) + np.float32(127)
    sd_bits = q.astype(np.uint64)
    keys, block_of = np.unique(mc & ~np.uint64(63), return_inverse=True)
    tsdf = np.full((len(keys), 64), 0xFF, dtype=np.uint64)
    wts = np.full((len(keys), 64), 0xFF, dtype=np.uint64)
    cell = (mc & np.uint64(63)).astype(np.int64)
    tsdf[block_of, cell] = sd_bits
    wts[block_of, cell] = np.minimum(w, 254).astype(np.uint64)
    shifts = (np.arange(8, dtype=np.uint64) * np.uint64(8))
    pack = lambda a: np.bitwise_or.reduce(a.reshape(len(keys), 8, 8) << shifts, axis=2)
    return keys, np.ascontiguousarray(pack(tsdf)), np.ascontiguousarray(pack(wts)) if saturate else None

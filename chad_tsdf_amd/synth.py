"""Synthetic depth scans for the BASELINE.json configurations.

Bit-reproducible across numpy / C++ / HIP: a splitmix64 stream, u01 = (x >> 11) * 2^-53 in f64,
all geometry in f64 and one final cast to f32.  No trigonometry (libm differs across hosts): ray
directions are three uniforms in [-1, 1], rejected unless 1e-3 <= |d| <= 1, then normalised.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, count: int, offset: int = 0) -> np.ndarray:
    """Outputs offset .. offset+count-1 of the splitmix64 stream started at `seed` (counter form)."""
    with np.errstate(over="ignore"):
        k = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        return z ^ (z >> np.uint64(31))


def u01(seed: int, count: int, offset: int = 0) -> np.ndarray:
    return (splitmix64(seed, count, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def ray_directions(seed: int, n: int) -> np.ndarray:
    """First n accepted unit directions (f64, [n,3]) of the rejection sampler over the stream."""
    out = np.empty((0, 3), dtype=np.float64)
    offset = 0
    while out.shape[0] < n:
        m = max(1024, int((n - out.shape[0]) * 2.2) + 64)
        d = 2.0 * u01(seed, 3 * m, offset).reshape(m, 3) - 1.0
        offset += 3 * m
        r = np.sqrt((d * d).sum(axis=1))
        keep = (r >= 1e-3) & (r <= 1.0)
        out = np.concatenate([out, d[keep] / r[keep, None]], axis=0)
    return out[:n]


def box_scan(n: int, lo, hi, sensor, seed: int) -> np.ndarray:
    """n noise-free hits of rays from `sensor` (strictly inside) on the axis-aligned box [lo, hi]; f32 [n,3]."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    o = np.asarray(sensor, dtype=np.float64)
    d = ray_directions(seed, n)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(d > 0, (hi - o) / d, np.where(d < 0, (lo - o) / d, np.inf))
    tmin = t.min(axis=1)
    return (o + tmin[:, None] * d).astype(np.float32)


def plane_scan(ny: int = 400, nz: int = 250, x: float = 4.03, seed: int = 1) -> np.ndarray:
    """cfg1: jittered grid on the plane x = const, ny columns over y in [-4,4] by nz rows over z in [-2,2];
    row-major (z is the row), two draws per cell (y jitter, then z jitter); f32 [ny*nz,3]."""
    u = u01(seed, 2 * ny * nz).reshape(nz, ny, 2)
    iy = np.arange(ny, dtype=np.float64)[None, :]
    iz = np.arange(nz, dtype=np.float64)[:, None]
    y = -4.0 + 8.0 * (iy + u[:, :, 0]) / ny
    z = -2.0 + 4.0 * (iz + u[:, :, 1]) / nz
    pts = np.stack([np.full_like(y, x), y, z], axis=-1).reshape(-1, 3)
    return pts.astype(np.float32)


# ---- the five BASELINE.json configurations ----------------------------------------------------
CFG1 = dict(res=0.10, trunc=0.20, sensor=(0.0, 0.0, 0.0))
CFG2 = dict(res=0.05, trunc=0.10, n=1_000_000, lo=(-5.0, -4.0, -1.5), hi=(5.0, 4.0, 1.5), sensor=(0.3, -0.2, 0.1), seed=2)
CFG3 = dict(res=0.05, trunc=0.10, n=200_000, lo=(-15.0, -4.0, -1.5), hi=(15.0, 4.0, 1.5), frames=500)
CFG5 = dict(res=0.02, trunc=0.06, n=200_000, lo=(-20.0, -20.0, -1.5), hi=(20.0, 20.0, 10.0))


def cfg1_frame():
    return plane_scan(), np.asarray(CFG1["sensor"], dtype=np.float32)


def cfg2_frame(n: int | None = None, seed: int | None = None, sensor=None):
    c = CFG2
    s = c["sensor"] if sensor is None else sensor
    return box_scan(c["n"] if n is None else n, c["lo"], c["hi"], s, c["seed"] if seed is None else seed), np.asarray(s, dtype=np.float32)


def cfg3_frame(k: int, n: int | None = None):
    """Frame k of the corridor stream: sensor (-12 + 0.05 k, 0.1, 0.0), seed 1000 + k."""
    c = CFG3
    s = (-12.0 + 0.05 * k, 0.1, 0.0)
    return box_scan(c["n"] if n is None else n, c["lo"], c["hi"], s, 1000 + k), np.asarray(s, dtype=np.float32)


def cfg5_frame(k: int, n: int | None = None):
    """Frame k of the outdoor trajectory: the box slides with the sensor, 0.5 m per frame along x; seed 5000 + k."""
    c = CFG5
    p = np.array([0.5 * k, 0.0, 0.0])
    lo = np.asarray(c["lo"]) + p
    hi = np.asarray(c["hi"]) + p
    return box_scan(c["n"] if n is None else n, lo, hi, p, 5000 + k), p.astype(np.float32)


# ---- the same generators on the GPU (torch tensors; tests and bench.py only -- never part of the product path) ------------------
# A 5 000-frame trajectory is a billion points: numpy needs 0.2 s per frame and core, the device a millisecond.  Every operation
# below is one IEEE f64 (or wrapping int64) operation per element and torch op -- no fused multiply-add can form across ops -- so
# the frames come out bit for bit as box_scan() makes them (tests/test_gpu_streams.py checks that on the frames it hands the oracle).
def _i64(x: int) -> int:
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


def _splitmix64_torch(seed: int, count: int, offset: int, device):
    import torch

    def lsr(z, s):   # logical shift right of an int64
        return (z >> s) & ((1 << (64 - s)) - 1)

    k = torch.arange(offset + 1, offset + count + 1, dtype=torch.int64, device=device)
    z = k * _i64(0x9E3779B97F4A7C15) + _i64(seed)
    z = (z ^ lsr(z, 30)) * _i64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * _i64(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def box_scan_torch(n: int, lo, hi, sensor, seed: int, device="cuda"):
    """box_scan() on the GPU: f32 [n, 3] torch tensor, the same bits."""
    import torch

    f64 = dict(dtype=torch.float64, device=device)
    lo_t, hi_t, o = torch.tensor(lo, **f64), torch.tensor(hi, **f64), torch.tensor(sensor, **f64)
    parts, have, offset = [], 0, 0
    while have < n:
        m = max(1024, int((n - have) * 2.2) + 64)
        x = _splitmix64_torch(seed, 3 * m, offset, device)
        u = ((x >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / 9007199254740992.0)
        d = (2.0 * u).reshape(m, 3) - 1.0
        offset += 3 * m
        sq = d * d
        r = torch.sqrt((sq[:, 0] + sq[:, 1]) + sq[:, 2])
        keep = (r >= 1e-3) & (r <= 1.0)
        parts.append(d[keep] / r[keep, None])
        have += int(parts[-1].shape[0])
    d = torch.cat(parts, dim=0)[:n]
    inf = torch.full_like(d, float("inf"))
    t = torch.where(d > 0, (hi_t - o) / d, torch.where(d < 0, (lo_t - o) / d, inf))
    tmin = t.min(dim=1).values
    return (o + tmin[:, None] * d).to(torch.float32)


def cfg5_frame_torch(k: int, n: int | None = None, device="cuda"):
    """cfg5_frame(k) made on the device: (points f32 [n, 3] CUDA tensor, pose f32 numpy [3])."""
    c = CFG5
    p = np.array([0.5 * k, 0.0, 0.0])
    lo = (np.asarray(c["lo"]) + p).tolist()
    hi = (np.asarray(c["hi"]) + p).tolist()
    return box_scan_torch(c["n"] if n is None else n, lo, hi, p.tolist(), 5000 + k, device), p.astype(np.float32)


def cfg3_frame_torch(k: int, n: int | None = None, device="cuda"):
    """cfg3_frame(k) made on the device."""
    c = CFG3
    sensor = (-12.0 + 0.05 * k, 0.1, 0.0)
    return box_scan_torch(c["n"] if n is None else n, list(c["lo"]), list(c["hi"]), list(sensor), 1000 + k, device), np.asarray(sensor, dtype=np.float32)

#!/usr/bin/env python3
"""Host time of reset() and insert() in the bench loop (three frames in flight), and the step time with and without the reset."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth
scans = [synth.cfg2_frame(1_000_000, seed=2 + f, sensor=(0.3 + 0.01 * f, -0.2, 0.1)) for f in range(4)]
dev = [torch.from_numpy(p).cuda() for p, _ in scans]
torch.cuda.synchronize()
for with_reset in (True, False):
    m = TSDFMap(0.05, 0.10, max_points=1_000_000)
    for i in range(6):
        m.reset(); m.insert(dev[i % 4], scans[i % 4][1], ready=True)
    m.sync()
    tr = ti = 0.0
    K = 400
    t0 = time.perf_counter()
    for i in range(K):
        a = time.perf_counter()
        if with_reset: m.reset()
        b = time.perf_counter()
        m.insert(dev[i % 4], scans[i % 4][1], ready=True)
        c = time.perf_counter()
        tr += b - a; ti += c - b
    m.sync()
    t = time.perf_counter() - t0
    print(f"reset={with_reset}: {t/K*1e6:.1f} us per frame; host: reset {tr/K*1e6:.1f} us, insert {ti/K*1e6:.1f} us (incl. waiting for the device's consumed word)")
    m.close()
os.environ["CHAD_DEBUG_NO_THROTTLE"] = "1"
m = TSDFMap(0.05, 0.10, max_points=1_000_000)
for i in range(6):
    m.reset(); m.insert(dev[i % 4], scans[i % 4][1], ready=True)
m.sync()
tr = ti = 0.0
K = 200
for i in range(K):
    a = time.perf_counter(); m.reset(); b = time.perf_counter(); m.insert(dev[i % 4], scans[i % 4][1], ready=True); c = time.perf_counter()
    tr += b - a; ti += c - b
m.sync()
print(f"no throttle (pure host cost): reset {tr/K*1e6:.1f} us, insert {ti/K*1e6:.1f} us")

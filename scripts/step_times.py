#!/usr/bin/env python3
"""Host time of every reset + insert of the bench loop: where a fixed cost inside the timed region hides."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth
scans = [synth.cfg2_frame(1_000_000, seed=2 + f, sensor=(0.3 + 0.01 * f, -0.2, 0.1)) for f in range(4)]
dev = [torch.from_numpy(p).cuda() for p, _ in scans]
torch.cuda.synchronize()
m = TSDFMap(0.05, 0.10, time_update=True, max_points=1_000_000)
for i in range(3):
    m.reset(); m.insert(dev[i % 4], scans[i % 4][1], ready=True)
m.sync(); m.stats()
torch.cuda.synchronize()
for K in (25, 50, 100, 200, 400):
    ts = []
    t0 = time.perf_counter()
    for i in range(K):
        t1 = time.perf_counter()
        m.reset(); m.insert(dev[i % 4], scans[i % 4][1], ready=True)
        ts.append(time.perf_counter() - t1)
    t_enq = time.perf_counter() - t0
    m.sync(); torch.cuda.synchronize()
    t = time.perf_counter() - t0
    ts = np.array(ts) * 1e6
    print(f"K={K}: total {t*1e3:.3f} ms = {t/K*1e6:.1f} us/step; enqueue {t_enq*1e3:.3f} ms; drain {(t-t_enq)*1e3:.3f} ms; step host us: median {np.median(ts):.1f} max {ts.max():.1f} at {ts.argmax()} first5 {np.round(ts[:5],1)}")

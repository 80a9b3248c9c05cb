#!/usr/bin/env python3
"""Per-kernel durations (dispatch time stamps) one frame at a time against three frames in flight (CHAD_DEBUG_TIME_PIPELINED): which
kernels stretch when they share the chip."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scans = [synth.cfg2_frame(n, seed=2 + f, sensor=(0.3 + 0.01 * f, -0.2, 0.1)) for f in range(4)]
dev = [torch.from_numpy(p).cuda() for p, _ in scans]
torch.cuda.synchronize()
out = {}
for mode in ("serial", "pipelined"):
    if mode == "pipelined":
        os.environ["CHAD_DEBUG_TIME_PIPELINED"] = "1"
    m = TSDFMap(0.05, 0.10, time_kernels=True, max_points=n)
    for i in range(6):
        m.reset(); m.insert(dev[i % 4], scans[i % 4][1], ready=True)
    m.kernel_times()
    t0 = time.perf_counter()
    for rep in range(5):          # the timing ring holds 16 inserts
        for i in range(12):
            m.reset(); m.insert(dev[i % 4], scans[i % 4][1], ready=True)
        m.sync()
    dt = (time.perf_counter() - t0) / 60
    kt = m.kernel_times()
    out[mode] = {"us_per_step": round(dt * 1e6, 1), "kernel_us": {k: round(v[0] * 1e3, 1) for k, v in kt.items()}, "sum_us": round(sum(v[0] for v in kt.values()) * 1e3, 1)}
    m.close()
print(json.dumps(out, indent=1))

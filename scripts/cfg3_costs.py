#!/usr/bin/env python3
"""cfg3-sized frames (200 k points): host enqueue cost per insert (map far from full: nothing throttles the host for the first
frames of a burst) and the serial per-kernel device times."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth

frames = [synth.cfg3_frame(k) for k in range(40)]
dev = [torch.from_numpy(p).cuda() for p, _ in frames]
torch.cuda.synchronize()
out = {}
m = TSDFMap(0.05, 0.10, max_points=200_000)
for (_, pos), d in zip(frames[:6], dev[:6]):
    m.insert(d, pos, ready=True)
m.sync()
# host cost of an insert while the device keeps up: insert, wait, insert, wait ...
t_sum = 0.0
for (_, pos), d in zip(frames[6:], dev[6:]):
    t0 = time.perf_counter(); m.insert(d, pos, ready=True); t_sum += time.perf_counter() - t0
    m.sync()
out["host_us_per_insert_ready"] = round(t_sum / 34 * 1e6, 1)
t_sum = 0.0
for (_, pos), d in zip(frames[6:], dev[6:]):
    t0 = time.perf_counter(); m.insert(d, pos); t_sum += time.perf_counter() - t0
    m.sync()
out["host_us_per_insert_stream_ordered"] = round(t_sum / 34 * 1e6, 1)
# back to back (three frames in flight; the host waits for the device's consumed word when it runs ahead)
m.reset(); m.sync()
t0 = time.perf_counter()
for rep in range(5):
    for (_, pos), d in zip(frames, dev):
        m.insert(d, pos, ready=True)
t_host = time.perf_counter() - t0
m.sync()
t_all = time.perf_counter() - t0
out["back_to_back_us_per_frame"] = round(t_all / 200 * 1e6, 1); out["back_to_back_host_us_per_frame"] = round(t_host / 200 * 1e6, 1)
m.close()
mk = TSDFMap(0.05, 0.10, max_points=200_000, time_kernels=True)
for (_, pos), d in zip(frames, dev):
    mk.insert(d, pos, ready=True)
kt = mk.kernel_times()
out["kernel_us_serial"] = {k: round(v[0] * 1e3, 1) for k, v in kt.items()}
out["kernel_us_sum_serial"] = round(sum(v[0] for v in kt.values()) * 1e3, 1)
mk.close()
print(json.dumps(out))

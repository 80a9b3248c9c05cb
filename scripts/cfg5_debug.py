import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth
frames = [synth.cfg5_frame(k) for k in range(33)]
dev = [torch.from_numpy(p).cuda() for p, _ in frames]
torch.cuda.synchronize()
m5 = TSDFMap(0.02, 0.06, time_update=True, max_points=200_000)
m5.insert(dev[0], frames[0][1], ready=True)
v5 = len(m5.export_leaves()[0])
m5.reset(); m5.sync(); m5.stats()
ts = []
t1 = time.perf_counter()
for k in range(33):
    a = time.perf_counter()
    m5.insert(dev[k], frames[k][1], ready=True)
    ts.append((time.perf_counter() - a) * 1e3)
m5.sync()
print("total ms", (time.perf_counter() - t1) * 1e3, "per-insert host ms", np.round(ts, 2).tolist())
print(m5.stats())

#!/usr/bin/env python3
"""cfg5-shaped frames (200 k points, 0.02 m voxels): per-kernel device times one frame at a time, and the roll-over's cost."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth
frames = [synth.cfg5_frame(k) for k in range(24)]
dev = [torch.from_numpy(p).cuda() for p, _ in frames]
torch.cuda.synchronize()
mk = TSDFMap(0.02, 0.06, max_points=200_000, time_kernels=True)
for (_, pos), d in list(zip(frames, dev))[:4]:
    mk.insert(d, pos, ready=True)
mk.kernel_times()
t0 = time.perf_counter()
for (_, pos), d in list(zip(frames, dev))[4:]:
    mk.insert(d, pos, ready=True)
mk.sync()
dt = (time.perf_counter() - t0) / 20
kt = mk.kernel_times()
st = mk.stats()
print(json.dumps({"ms_per_frame_serial": round(dt * 1e3, 4), "kernel_us": {k: round(v[0] * 1e3, 1) for k, v in kt.items()}, "sum_us": round(sum(v[0] for v in kt.values()) * 1e3, 1),
                  "sort_passes": st["sort_passes"], "blocks": st["blocks"], "closed": st["closed_submaps"]}))

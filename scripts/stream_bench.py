#!/usr/bin/env python3
"""cfg3 of BASELINE.json: a stream of 200 k-point frames from a moving sensor (0.05 m per frame, roll-over every 101 frames),
device-resident, inserted back to back.  Prints steady-state throughput and the hash occupancy; --cpu-frames N also times
the oracle (CPU restatement) on the first N frames.  Not the bench line (bench.py is): a measurement script for DESIGN.md."""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chad_tsdf_amd import TSDFMap, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=500)
ap.add_argument("--cpu-frames", type=int, default=0)
args = ap.parse_args()
frames = [synth.cfg3_frame(k) for k in range(args.frames)]
dev = [torch.from_numpy(p).cuda() for p, _ in frames]
torch.cuda.synchronize()
m = TSDFMap(0.05, 0.10)
for (_, pos), d in list(zip(frames, dev))[:5]:
    m.insert(d, pos)
m.sync(); m.reset(); m.sync()
t0 = time.perf_counter()
for (_, pos), d in zip(frames, dev):
    m.insert(d, pos)
t_host = time.perf_counter() - t0
m.sync()
t = time.perf_counter() - t0
st = m.stats()
out = {"workload": "cfg3: %d frames x 200k points, voxel 0.05 m, trunc 0.10 m, sensor moving 0.05 m/frame" % args.frames,
       "ms_per_frame": t / args.frames * 1e3, "points_per_s": args.frames * 200_000 / t, "host_enqueue_ms_per_frame": t_host / args.frames * 1e3,
       "closed_submaps": st["closed_submaps"], "live_blocks": st["blocks"], "block_capacity": st["block_capacity"],
       "hash_load": st["blocks"] / st["hash_capacity"], "overflow": st["overflow"]}
if args.cpu_frames:
    from oracle import pyoracle
    o = pyoracle.OracleMap(0.05, 0.10)
    t1 = time.perf_counter()
    for pts, pos in frames[: args.cpu_frames]:
        o.insert(pts, pos)
    tc = time.perf_counter() - t1
    out["cpu_ms_per_frame"] = tc / args.cpu_frames * 1e3
    out["speedup_vs_cpu"] = out["cpu_ms_per_frame"] / out["ms_per_frame"]
print(json.dumps(out))

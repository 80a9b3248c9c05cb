#!/usr/bin/env python3
"""cfg5 of BASELINE.json on one GPU: outdoor trajectory, 200 k points per frame, voxel 0.02 m, trunc 0.06 m, sensor advancing
0.5 m per frame (a roll-over every 11 frames).  Prints throughput, live blocks and the HBM held by the map."""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chad_tsdf_amd import TSDFMap, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=100)
ap.add_argument("--update-mode", type=int, default=0)
args = ap.parse_args()
frames = [synth.cfg5_frame(k) for k in range(args.frames)]
dev = [torch.from_numpy(p).cuda() for p, _ in frames]
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
m = TSDFMap(0.02, 0.06, update_mode=args.update_mode)
for (_, pos), d in list(zip(frames, dev))[:3]:
    m.insert(d, pos)
m.sync(); m.reset(); m.sync()
t0 = time.perf_counter()
for (_, pos), d in zip(frames, dev):
    m.insert(d, pos)
t_host = time.perf_counter() - t0
m.sync()
t = time.perf_counter() - t0
st = m.stats()
out = {"workload": "cfg5: %d frames x 200k points, voxel 0.02 m, trunc 0.06 m, 0.5 m per frame" % args.frames, "update_mode": args.update_mode,
       "ms_per_frame": round(t / args.frames * 1e3, 4), "points_per_s": round(args.frames * 200_000 / t), "host_enqueue_ms_per_frame": round(t_host / args.frames * 1e3, 4),
       "updates_last_frame": st["updates"], "closed_submaps": st["closed_submaps"], "live_blocks": st["blocks"], "block_capacity": st["block_capacity"],
       "hash_load": round(st["blocks"] / st["hash_capacity"], 4), "overflow": st["overflow"],
       "hbm_held_GiB": round((free0 - torch.cuda.mem_get_info()[0]) / 2**30, 2)}
print(json.dumps(out))

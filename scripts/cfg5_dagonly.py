#!/usr/bin/env python3
"""BASELINE config 5 under CHAD_FLAG_DAG_ONLY (the reference's policy: a closed submap is its DAG and nothing else): frames made on the
device, inserted back to back; prints the frame time, what finalize cost per submap, memory held.   --frames N (default 110)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.environ.get("CHAD_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (CHAD_AB_ROOT: another build's tree, scripts/ab_cfg5.sh)
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=110)
ap.add_argument("--keep", action="store_true", help="keep the dense archives (no DAG_ONLY)")
ap.add_argument("--alone", action="store_true", help="wait for every roll-over's fold before the next frame (its kernels' own times, for a trace)")
args = ap.parse_args()
dev = [synth.cfg5_frame_torch(k) for k in range(args.frames)]
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
m = TSDFMap(0.02, 0.06, max_points=200_000, max_blocks=1 << 22, dag_only=not args.keep)
for k in range(12):   # warm-up: kernels, the spare table, the DAG's first allocations
    m.insert(dev[k][0], dev[k][1], ready=True)
m.sync(); m.reset(); m.sync()
t0 = time.perf_counter()
for k in range(args.frames):
    m.insert(dev[k][0], dev[k][1], ready=True)
    if args.alone and k and k % 11 == 0:
        m.sync(); m.finalize()
t_host = time.perf_counter() - t0
m.sync()
t = time.perf_counter() - t0
t1 = time.perf_counter()
m.finalize()
t_fin = time.perf_counter() - t1
st, ds = m.stats(), m.dag_stats()
print(json.dumps({"frames": args.frames, "dag_only": not args.keep, "ms_per_frame": round(t / args.frames * 1e3, 4), "host_ms_per_frame": round(t_host / args.frames * 1e3, 4),
                  "closed": st["closed_submaps"], "finalize_after_ms": round(t_fin * 1e3, 2), "dag_ms_device_last": round(ds["ms_device"], 3), "dag_ms_host_last": round(ds["ms_host"], 3),
                  "unique_clusters": ds["uniques"][20], "hbm_held_GiB": round((free0 - torch.cuda.mem_get_info()[0]) / 2**30, 2), "overflow": st["overflow"]}))
m.close()

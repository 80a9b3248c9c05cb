#!/usr/bin/env python3
"""The sparse exchange's kernels at BASELINE config 5's scale, for a kernel trace on ONE GPU and ONE process: a two-rank communicator over
the callback transport whose peer holds nothing (tests/test_gpu_exchange_faults.py: SilentPeerComm), so this rank integrates every second
frame of the trajectory, runs all five stages of every roll-over's exchange (k_table_keys, the key sort, k_xchg_masks / prefix / pack /
accumulate / clusters / compact_rows / expand_rows) and, being rank 0, folds.   --frames N (default 110)
  rocprofv3 --kernel-trace --stats -d out -- python3 scripts/exchange_trace.py"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth
from tests.test_gpu_exchange_faults import SilentPeerComm

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=110)
args = ap.parse_args()
dev = {k: synth.cfg5_frame_torch(k)[0] for k in range(0, args.frames, 2)}
torch.cuda.synchronize()
comm = SilentPeerComm()
m = TSDFMap(0.02, 0.06, max_points=200_000, max_blocks=1 << 22, dag_only=True)
m.attach_comm(comm)
m.pipeline_times()
t0 = time.perf_counter()
for k in range(args.frames):
    pos = np.array([0.5 * k, 0.0, 0.0], dtype=np.float32)
    if k in dev:
        m.insert(dev[k], pos, ready=True)
    else:
        m.note_pose(pos)
m.close_submap()
m.sync()
t = time.perf_counter() - t0
pt = m.pipeline_times()
print(json.dumps({"frames": args.frames, "integrated": len(dev), "ms_per_frame_of_the_stream": round(t / args.frames * 1e3, 4), "exchanges": pt["exchanges"],
                  "MB_to_the_exchange_per_rollover": round(pt["exchange_bytes"] / max(1, pt["exchanges"]) / 1e6, 2), "union_blocks_last": pt["exchange_union_blocks"],
                  "folds": pt["folds_completed"], "step_a_ms": round(pt["step_a_ms"] / max(1, pt["folds_completed"]), 3), "step_b_ms": round(pt["step_b_ms"] / max(1, pt["folds_completed"]), 3)}))
m.attach_comm(None)
m.close(); comm.close()

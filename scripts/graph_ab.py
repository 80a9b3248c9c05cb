#!/usr/bin/env python3
"""The preparation chain launched kernel by kernel against the same chain replayed as a hipGraph (frame_graph.hpp): streams of small
frames (cfg3's 200 k points, and 20 k points), and the headline's reset + insert of 1 M-point frames.  One child process per setting."""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(kind):
    import numpy as np, torch
    from chad_tsdf_amd import TSDFMap, synth
    if kind == "cfg2":
        scans = [synth.cfg2_frame(1_000_000, seed=2 + f, sensor=(0.3 + 0.01 * f, -0.2, 0.1)) for f in range(4)]
        dev = [torch.from_numpy(p).cuda() for p, _ in scans]
        torch.cuda.synchronize()
        m = TSDFMap(0.05, 0.10, max_points=1_000_000)
        for i in range(12):
            m.reset(); m.insert(dev[i % 4], scans[i % 4][1], ready=True)
        m.sync(); m.host_times()
        t0 = time.perf_counter()
        for i in range(400):
            m.reset(); m.insert(dev[i % 4], scans[i % 4][1], ready=True)
        m.sync()
        t = (time.perf_counter() - t0) / 400
    else:
        n = 200_000 if kind == "cfg3" else 20_000
        frames = [synth.cfg3_frame(k, n) for k in range(500)]
        dev = [torch.from_numpy(p).cuda() for p, _ in frames]
        torch.cuda.synchronize()
        m = TSDFMap(0.05, 0.10, max_points=n)
        for k in range(12):
            m.insert(dev[k], frames[k][1], ready=True)
        m.sync(); m.reset(); m.sync(); m.host_times()
        t0 = time.perf_counter()
        for k in range(500):
            m.insert(dev[k], frames[k][1], ready=True)
        m.sync()
        t = (time.perf_counter() - t0) / 500
    h = m.host_times()
    print(json.dumps({"kind": kind, "graphs": os.environ.get("CHAD_GRAPHS", "1"), "slots": os.environ.get("CHAD_SLOTS", "3"), "hwq": os.environ.get("GPU_MAX_HW_QUEUES", "-"),
                      "ms_per_frame": round(t * 1e3, 4), "host_us": {k: round(v, 1) for k, v in h.items() if k != "inserts"}}), flush=True)
    m.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2]); sys.exit(0)
    for kind in ("cfg3", "20k", "cfg2"):
        for env in ({"CHAD_GRAPHS": "0"}, {"CHAD_GRAPHS": "1"}, {"CHAD_GRAPHS": "1", "CHAD_SLOTS": "4", "GPU_MAX_HW_QUEUES": "8"}, {"CHAD_GRAPHS": "0", "CHAD_SLOTS": "4", "GPU_MAX_HW_QUEUES": "8"}):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", kind], env=dict(os.environ, **env), capture_output=True, text=True)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-600:], flush=True)

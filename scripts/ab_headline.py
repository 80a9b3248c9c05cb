#!/usr/bin/env python3
"""The headline step (reset + insert of a 1 M-point cfg2 frame, three frames in flight) and the cfg3 stream with this tree's library
against another build of it in <dir> (e.g. a checkout of an earlier commit under scratch/), alternating on the same box.
    python3 scripts/ab_headline.py scratch/r02"""
import json, os, subprocess, sys, time


def child(root):
    sys.path.insert(0, root)
    import numpy as np, torch
    from chad_tsdf_amd import TSDFMap, synth
    scans = [synth.cfg2_frame(1_000_000, seed=2 + f, sensor=(0.3 + 0.01 * f, -0.2, 0.1)) for f in range(4)]
    dev = [torch.from_numpy(p).cuda() for p, _ in scans]
    torch.cuda.synchronize()
    m = TSDFMap(0.05, 0.10, max_points=1_000_000)
    for i in range(12):
        m.reset(); m.insert(dev[i % 4], scans[i % 4][1], ready=True)
    m.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(400):
            m.reset(); m.insert(dev[i % 4], scans[i % 4][1], ready=True)
        m.sync()
        best = min(best, (time.perf_counter() - t0) / 400)
    m.close()
    frames = [synth.cfg3_frame(k) for k in range(300)]
    dev = [torch.from_numpy(p).cuda() for p, _ in frames]
    torch.cuda.synchronize()
    m = TSDFMap(0.05, 0.10, max_points=200_000)
    for k in range(12):
        m.insert(dev[k], frames[k][1], ready=True)
    m.sync(); m.reset(); m.sync()
    t0 = time.perf_counter()
    for k in range(300):
        m.insert(dev[k], frames[k][1], ready=True)
    m.sync()
    t3 = (time.perf_counter() - t0) / 300
    m.close()
    # cfg5's shape: the same eleven frames over and over (no roll-over), then 66 frames with roll-overs
    frames = [synth.cfg5_frame(k) for k in range(66)]
    dev = [torch.from_numpy(p).cuda() for p, _ in frames]
    torch.cuda.synchronize()
    out5 = {}
    for mode in ("noroll", "roll"):
        m = TSDFMap(0.02, 0.06, max_points=200_000, max_blocks=1 << 22)
        for k in range(3):
            m.insert(dev[k], frames[k][1], ready=True)
        m.close_submap(); m.sync(); m.reset(); m.sync()
        if mode == "noroll":
            for k in range(11):
                m.insert(dev[k], frames[k][1], ready=True)
            m.sync()
        t0 = time.perf_counter()
        for k in range(66):
            j = k % 11 if mode == "noroll" else k
            m.insert(dev[j], frames[j][1], ready=True)
        m.sync()
        out5[mode] = round((time.perf_counter() - t0) / 66 * 1e3, 4)
        m.close()
    print(json.dumps({"root": root, "cfg2_ms_per_frame": round(best * 1e3, 4), "cfg3_ms_per_frame": round(t3 * 1e3, 4), "cfg5_noroll": out5["noroll"], "cfg5_roll": out5["roll"]}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2]); sys.exit(0)
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    other = os.path.abspath(sys.argv[1])
    for rep in range(3):
        for root in (here, other):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", root], capture_output=True, text=True, cwd=root)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-500:], flush=True)

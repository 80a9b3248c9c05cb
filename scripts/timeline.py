#!/usr/bin/env python3
"""Frames in flight as the device saw them: every kernel's dispatch time stamps (CHAD_DEBUG_TIME_PIPELINED + CHAD_DEBUG_TIMELINE, no
profiler on the host's back) for a dozen 1 M-point frames -- how many kernels run at any time, what each frame's stream waits for."""
import collections, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from chad_tsdf_amd import TSDFMap, synth
    n = int(sys.argv[2])
    with_reset = len(sys.argv) < 4 or sys.argv[3] != "noreset"
    scans = [synth.cfg2_frame(n, seed=2 + f, sensor=(0.3 + 0.01 * f, -0.2, 0.1)) for f in range(4)]
    dev = [torch.from_numpy(p).cuda() for p, _ in scans]
    torch.cuda.synchronize()
    m = TSDFMap(0.05, 0.10, time_kernels=True, max_points=n)
    for i in range(6):
        m.reset(); m.insert(dev[i % 4], scans[i % 4][1], ready=True)
    m.kernel_times()
    for i in range(14):
        if with_reset: m.reset()
        m.insert(dev[i % 4], scans[i % 4][1], ready=True)
    m.sync(); m.kernel_times(); m.close()
    sys.exit(0)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
env = dict(os.environ, CHAD_DEBUG_TIME_PIPELINED="1", CHAD_DEBUG_TIMELINE="1")
err = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n)] + sys.argv[2:3], env=env, capture_output=True, text=True).stderr
names = ["bounds", "keys"] + [f"sort{p}" for p in range(8)] + ["gather", "normals", "update"]
ev = []
for line in err.splitlines():
    if line.startswith("timeline frame"):
        w = line.split()
        ev.append((int(w[2]), names[int(w[4])], float(w[6]), float(w[8])))
frames = sorted({e[0] for e in ev})
frames = frames[-12:-1]                      # steady state: the last dozen, without the very last (the pipeline drains behind it)
sel = [e for e in ev if e[0] in frames]
t0 = min(e[2] for e in sel); t1 = max(e[3] for e in sel)
pts = sorted([(e[2], 1) for e in sel] + [(e[3], -1) for e in sel])
hist = collections.Counter(); depth = 0; last = t0
for t, d in pts:
    hist[depth] += t - last; depth += d; last = t
starts = sorted(min(e[2] for e in sel if e[0] == f) for f in frames)
print(f"{len(frames)} frames, {n} points: a frame every {(starts[-1] - starts[0]) / (len(starts) - 1):.1f} us; kernels running at a time: " +
      ", ".join(f"{k}: {100 * v / (t1 - t0):.0f} %" for k, v in sorted(hist.items())))
life = [max(e[3] for e in sel if e[0] == f) - min(e[2] for e in sel if e[0] == f) for f in frames]
busy = [sum(e[3] - e[2] for e in sel if e[0] == f) for f in frames]
print(f"a frame lives {sum(life) / len(life):.1f} us on its stream, {sum(busy) / len(busy):.1f} us of it inside kernels")
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for f in frames:
    mine = sorted([e for e in sel if e[0] == f], key=lambda e: e[2])
    for a, b in zip(mine, mine[1:]): gap[f"{a[1]}->{b[1]}"].append(b[2] - a[3])
    for e in mine: dur[e[1]].append(e[3] - e[2])
print("kernel us in flight: " + ", ".join(f"{k} {sum(v) / len(v):.1f}" for k, v in dur.items()))
print("gaps us on the frame's stream: " + ", ".join(f"{k} {sum(v) / len(v):.1f}" for k, v in gap.items()))

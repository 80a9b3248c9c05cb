#!/usr/bin/env python3
"""cfg5's shape (200 k points per frame, 0.02 m voxels) taken apart: what a frame costs with and without the roll-over's background
work (archive compaction, clear), with 3 / 4 / 6 frame slots, and the frames in flight as the device sees them (dispatch time stamps).
    python3 scripts/cfg5_probe.py            all legs, one child process each (environment knobs differ)
"""
import collections, json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(mode, frames_n):
    import numpy as np, torch
    from chad_tsdf_amd import TSDFMap, synth
    frames = [synth.cfg5_frame(k) for k in range(frames_n if "noroll" not in mode else 11)]
    dev = [torch.from_numpy(p).cuda() for p, _ in frames]
    torch.cuda.synchronize()
    timeline = "timeline" in mode
    m = TSDFMap(0.02, 0.06, max_points=200_000, max_blocks=1 << 22, time_kernels=timeline, dag_only=(mode == "dagonly"))
    # warm-up: first frames, one roll-over (the spare table), then an empty map
    for k in range(3):
        m.insert(dev[k], frames[k][1], ready=True)
    m.close_submap(); m.sync(); m.reset(); m.sync()
    if timeline:
        m.kernel_times()
    if "noroll" in mode:   # the same eleven frames over and over: never more than 5 m from the first pose, every block exists after the first round
        for k in range(11):
            m.insert(dev[k], frames[k][1], ready=True)
        m.sync()
    t0 = time.perf_counter()
    for k in range(frames_n):
        j = k % len(dev)
        m.insert(dev[j], frames[j][1], ready=True)
    t_host = time.perf_counter() - t0
    m.sync()
    t = time.perf_counter() - t0
    st = m.stats()
    print(json.dumps({"mode": mode, "slots": os.environ.get("CHAD_SLOTS", "3"), "hwq": os.environ.get("GPU_MAX_HW_QUEUES", "-"), "frames": frames_n, "ms_per_frame": round(t / frames_n * 1e3, 4),
                      "host_ms_per_frame": round(t_host / frames_n * 1e3, 4), "closed": st["closed_submaps"], "blocks": st["blocks"], "overflow": st["overflow"]}), flush=True)
    if timeline:
        m.kernel_times()
    m.close()


def timeline_report(err, label):
    names = ["bounds", "keys"] + [f"sort{p}" for p in range(8)] + ["gather", "normals", "update"]
    ev = []
    for line in err.splitlines():
        if line.startswith("timeline frame"):
            w = line.split()
            ev.append((int(w[2]), names[int(w[4])], float(w[6]), float(w[8])))
    frames = sorted({e[0] for e in ev})[-24:-1]
    sel = [e for e in ev if e[0] in frames]
    if not sel:
        print(label, "no timeline"); return
    t0 = min(e[2] for e in sel); t1 = max(e[3] for e in sel)
    pts = sorted([(e[2], 1) for e in sel] + [(e[3], -1) for e in sel])
    hist = collections.Counter(); depth = 0; last = t0
    for t, d in pts:
        hist[depth] += t - last; depth += d; last = t
    starts = sorted(min(e[2] for e in sel if e[0] == f) for f in frames)
    print(f"{label}: {len(frames)} frames: a frame every {(starts[-1] - starts[0]) / (len(starts) - 1):.1f} us; kernels running at a time: " +
          ", ".join(f"{k}: {100 * v / (t1 - t0):.0f} %" for k, v in sorted(hist.items())))
    life = [max(e[3] for e in sel if e[0] == f) - min(e[2] for e in sel if e[0] == f) for f in frames]
    busy = [sum(e[3] - e[2] for e in sel if e[0] == f) for f in frames]
    print(f"  a frame lives {sum(life) / len(life):.1f} us on its stream, {sum(busy) / len(busy):.1f} us of it inside kernels")
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    for f in frames:
        mine = sorted([e for e in sel if e[0] == f], key=lambda e: e[2])
        for a, b in zip(mine, mine[1:]): gap[f"{a[1]}->{b[1]}"].append(b[2] - a[3])
        for e in mine: dur[e[1]].append(e[3] - e[2])
    print("  kernel us in flight: " + ", ".join(f"{k} {sum(v) / len(v):.1f}" for k, v in dur.items()))
    print("  gaps us: " + ", ".join(f"{k} {sum(v) / len(v):.1f}" for k, v in gap.items()))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    me = os.path.abspath(__file__)

    def run(mode, frames_n, **env):
        e = dict(os.environ, **{k: str(v) for k, v in env.items()})
        r = subprocess.run([sys.executable, me, "child", mode, str(frames_n)], env=e, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + r.stderr[-400:]), flush=True)
        return r

    run("rollover", 66)
    run("norollover", 66)
    run("dagonly", 44)
    for slots in (2, 4, 6):
        run("rollover", 66, CHAD_SLOTS=slots, GPU_MAX_HW_QUEUES=8)
        run("norollover", 66, CHAD_SLOTS=slots, GPU_MAX_HW_QUEUES=8)
    r = run("timeline", 33, CHAD_DEBUG_TIME_PIPELINED=1, CHAD_DEBUG_TIMELINE=1)
    timeline_report(r.stderr, "cfg5 with roll-overs, 3 slots")
    r = run("timeline_noroll", 33, CHAD_DEBUG_TIME_PIPELINED=1, CHAD_DEBUG_TIMELINE=1)
    timeline_report(r.stderr, "cfg5 without roll-overs, 3 slots")

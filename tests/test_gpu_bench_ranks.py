"""bench.py with two ranks: the N > 1 path of the contract's command (the driver starts it on an 8-GPU node; this build has only ever had
one GPU), rehearsed with two gloo ranks sharing the one GPU -- so that the cfg4 / cfg5_sharded legs and the strong-scaling field cannot
rot unseen.  The callback transport stands in for RCCL here (RCCL refuses two ranks on one GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_with_two_gloo_ranks_prints_the_strong_scaling_legs():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--points", "200000",
           "--stream-frames", "30", "--sharded-outdoor-frames", "22", "--cpu-seconds", "0", "--leg-timeout", "240"]
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    errors = "\n".join(l for l in done.stderr.splitlines() if not l.startswith(("W1", "W0", "I1", "I0")))   # (without the launcher's own chatter)
    assert done.returncode == 0, errors[:3000] + "\n...\n" + errors[-1500:]
    lines = [l for l in done.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and "watchdog" not in line
    cfg4 = line["cfg4"]
    assert "error" not in cfg4 and cfg4["scaling"] == "strong" and cfg4["n_gpus"] == 2 and cfg4["speedup_vs_1gpu"] > 0
    sharded = line["cfg5_sharded"]
    assert "error" not in sharded and sharded["speedup_vs_1gpu"] > 0
    strong = line["strong_scaling"]
    assert strong["n_gpus"] == 2 and strong["speedup_vs_1gpu"] == cfg4["speedup_vs_1gpu"]
    # the transport saw both ranks
    assert sharded["ranks_seen_by_transport"] == 2


@pytest.mark.gpu
def test_bench_watchdog_prints_the_contract_line_and_fails_loudly():
    """Extra legs that outlast --leg-timeout: the contract's line is still printed -- with what was measured until then and the leg that was
    running -- and the process ends with a non-zero code (3), so that a hung leg cannot pass for a clean run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--points", "200000", "--cpu-seconds", "0", "--leg-timeout", "0.05"]
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert done.returncode == 3, (done.returncode, done.stderr[-2000:])
    lines = [l for l in done.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["metric"] == "inserted points/sec" and line["value"] > 0 and line["n_gpus"] == 1 and line["steps"] == 2
    assert "watchdog" in line and line["watchdog_leg"]
    assert "roofline" in line

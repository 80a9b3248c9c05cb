#!/usr/bin/env python3
"""chad_map_insert with host points (the reference's signature): ms per 1 M-point frame against staging threads and chunk size."""
import os, subprocess, sys
code = r'''
import time, os, sys
sys.path.insert(0, ".")
from chad_tsdf_amd import TSDFMap, synth
scans = [synth.cfg2_frame(1_000_000, seed=2 + f) for f in range(4)]
m = TSDFMap(0.05, 0.10, max_points=1_000_000)
for i in range(4): m.reset(); m.insert(*scans[i % 4])
m.sync()
t0 = time.perf_counter()
for i in range(100): m.reset(); m.insert(*scans[i % 4])
m.sync()
print(round((time.perf_counter() - t0) * 10, 4))
'''
for threads in (1, 2, 3):
    for kb in (512, 1024, 2048, 3072, 6144):
        env = dict(os.environ, CHAD_STAGE_THREADS=str(threads), CHAD_STAGE_CHUNK_KB=str(kb))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print(f"threads {threads} chunk {kb} KiB: {out} ms per frame", flush=True)

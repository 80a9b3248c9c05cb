"""How much of the bench's step is the reset?  The same 1 M-point scans inserted back to back WITHOUT map.reset() (the map
keeps accumulating: every brick exists after the first pass over the scans) against the bench's reset + insert."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth

scans = []
for f in range(4):
    pts, pos = synth.cfg2_frame(1_000_000, seed=2 + f, sensor=(0.3 + 0.01 * f, -0.2, 0.1))
    scans.append((torch.from_numpy(pts).cuda(), pos))
for reset in (True, False):
    m = TSDFMap(0.05, 0.10)
    for i in range(8):
        m.insert(*scans[i % 4])
    m.sync()
    steps = 200
    t0 = time.perf_counter()
    for i in range(steps):
        if reset:
            m.reset()
        m.insert(*scans[i % 4])
    th = time.perf_counter() - t0
    m.sync()
    t = time.perf_counter() - t0
    print(f"slots {os.environ.get('CHAD_SLOTS', '3')} reset {reset}: {t / steps * 1e3:.4f} ms/step, host {th / steps * 1e3:.4f}")
    m.close()

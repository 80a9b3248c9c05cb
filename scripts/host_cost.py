import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth
pts, pos = synth.cfg3_frame(0, 20000)
d = torch.from_numpy(pts).cuda()
for kw in ({}, {"time_update": True}):
    m = TSDFMap(0.05, 0.10, **kw)
    for i in range(50): m.insert(d, pos)
    m.sync()
    t0 = time.perf_counter()
    N = 3000
    for i in range(N): m.insert(d, pos)
    t1 = time.perf_counter()
    m.sync()
    t2 = time.perf_counter()
    print(kw, "host us/insert %.1f  total us/insert %.1f" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
    m.close()
# the raw ctypes call cost
import ctypes as C
from chad_tsdf_amd import _lib
L = _lib.load()
m = TSDFMap(0.05, 0.10)
e = np.zeros((0, 3), np.float32)
t0 = time.perf_counter()
for i in range(20000): L.chad_map_insert_device(m._h, None, 0, pos.ctypes.data)
print("empty insert via ctypes: %.2f us" % ((time.perf_counter() - t0) / 20000 * 1e6))
t0 = time.perf_counter()
for i in range(20000): m.insert(e, pos)
print("empty insert via TSDFMap.insert: %.2f us" % ((time.perf_counter() - t0) / 20000 * 1e6))

#!/usr/bin/env python3
"""What one exchange (chad_map_merge_and_close) costs apart from the wire: a one-rank RCCL communicator that is not short-cut
(CHAD_DEBUG_COMM_NO_SHORTCUT), a cfg3 submap of 100 frames (23 k blocks), merged; against the plain local close."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CHAD_DEBUG_COMM_NO_SHORTCUT"] = "1"
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth, _lib

lib = _lib.load()
uid = (C.c_uint8 * 128)()
_lib.check(lib.chad_comm_unique_id(uid), "id")
comm = C.c_void_p()
_lib.check(lib.chad_comm_create_rccl(uid, 0, 1, C.byref(comm)), "comm")
class H: handle = comm
frames = [synth.cfg3_frame(k) for k in range(100)]
dev = [torch.from_numpy(p).cuda() for p, _ in frames]
torch.cuda.synchronize()
out = {}
for name in ("merge", "local_close"):
    m = TSDFMap(0.05, 0.10, max_points=200_000)
    for rep in range(4):   # the first close of a map also allocates and zeroes its spare table (3.3 GB at the default capacity): timed apart
        for (_, pos), d in zip(frames, dev):
            m.insert(d, pos, ready=True)
        m.sync()
        t0 = time.perf_counter()
        if name == "merge":
            m.merge_and_close(H)
        else:
            m.close_submap()
        m.sync()
        out.setdefault(name + "_ms", []).append(round((time.perf_counter() - t0) * 1e3, 3))
    out["blocks"] = m.block_count(0)
    m.close()
print(json.dumps(out))

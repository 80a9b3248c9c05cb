"""Host cost of the frame-sharded map per frame a rank does NOT own (pose bookkeeping only): what bounds the strong scaling
of a stream of small frames.  One process, no process group: the wrapper is told it is rank 1 of 1000."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chad_tsdf_amd.sharded import ShardedTSDFMap

for fast in (True, False):
    s = ShardedTSDFMap(0.05, 0.10)
    s.rank, s.world = 1, 1000
    s._fast = fast
    s.merge_and_close = lambda: (s.local.close_submap(), setattr(s, "_first_pose", None))   # no collective in this one-process measurement
    poses = [np.asarray((-12.0 + 0.01 * (k % 400), 0.1, 0.0), dtype=np.float32) for k in range(20000)]
    t0 = time.perf_counter()
    for k, pos in enumerate(poses):
        if k % 1000 != 1:
            s.insert(None, pos)
        else:
            s._frame += 1
    dt = (time.perf_counter() - t0) / len(poses) * 1e6
    print(f"non-owned frame, {'C calls' if fast else 'numpy path'}: {dt:.2f} us")

"""Submap::finalize: device + host time here against the oracle's finalize on one core.

  python tests/soak/finalize_bench.py            # cfg2 frame (9.4 k blocks) and a cfg3 submap (100 frames x 200 k points)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import chad_tsdf_amd as chad  # noqa: E402
from chad_tsdf_amd import synth  # noqa: E402
from oracle import pyoracle  # noqa: E402  (the CPU baseline beside the measurement, as in bench.py)


def run(name, frames, res, trunc):
    m = chad.TSDFMap(res, trunc)
    dev = torch.device("cuda", 0)
    for pts, pos in frames:
        m.insert(torch.from_numpy(pts).to(dev), pos)
    m.close_submap()
    m.sync()
    t0 = time.perf_counter()
    m.finalize()
    wall = (time.perf_counter() - t0) * 1e3
    st = m.dag_stats()
    blocks = m.block_count(0)
    leaves = m.export_leaves(0)
    o = pyoracle.OracleMap(res, trunc)
    o.close_submap_from_leaves(*leaves)
    same = all(np.array_equal(m.dag_level(l), o.dag_level(l)) for l in range(21))
    print(f"{name}: {blocks} blocks, {len(leaves[0])} leaves, {st['uniques'][20]} unique clusters: finalize wall {wall:.2f} ms "
          f"(device {st['ms_device']:.3f} ms incl. key sort and copies, host DAG {st['ms_host']:.3f} ms); oracle finalize {o.finalize_ms:.2f} ms on 1 core; "
          f"levels identical: {same}")
    # the cluster kernel alone: 520 B in + 72 B out per block
    m2 = chad.TSDFMap(res, trunc)
    for pts, pos in frames:
        m2.insert(torch.from_numpy(pts).to(dev), pos)
    m2.close_submap()
    m2.finalize()


if __name__ == "__main__":
    run("cfg2 frame", [synth.cfg2_frame()], 0.05, 0.10)
    run("cfg3 submap (100 frames)", [synth.cfg3_frame(k) for k in range(100)], 0.05, 0.10)

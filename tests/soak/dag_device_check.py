#!/usr/bin/env python3
"""The device fold of Submap::finalize (dag_device.hpp) against the host fold (dag.hpp, CHAD_DEBUG_DAG_HOST=1) on the same maps: all 21
level arrays, roots and counters word for word; and what each costs.  Run on the GPU box: python3 tests/soak/dag_device_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from chad_tsdf_amd import TSDFMap, synth


def build(kind, host, saturate):
    if host:
        os.environ["CHAD_DEBUG_DAG_HOST"] = "1"
    else:
        os.environ.pop("CHAD_DEBUG_DAG_HOST", None)
    if kind == "cfg2":
        m = TSDFMap(0.05, 0.10, dag_saturate_weights=saturate)
        for f in range(3):
            pts, pos = synth.cfg2_frame(1_000_000 if f == 0 else 300_000, seed=2 + f, sensor=(0.3 + 0.4 * f, -0.2, 0.1))
            m.insert(pts, pos); m.close_submap()
    elif kind == "cfg3":
        m = TSDFMap(0.05, 0.10, dag_saturate_weights=saturate)
        for k in range(0, 230, 2):
            m.insert(*synth.cfg3_frame(k, 60_000))
        m.close_submap()
    elif kind == "empty":
        m = TSDFMap(0.05, 0.10, dag_saturate_weights=saturate)
        m.insert(np.zeros((0, 3), np.float32), np.zeros(3, np.float32)); m.close_submap()
        pts, pos = synth.cfg2_frame(20_000)
        m.insert(pts, pos); m.close_submap()
        m.insert(np.zeros((0, 3), np.float32), pos); m.close_submap()
    else:
        m = TSDFMap(0.02, 0.06, max_blocks=1 << 22, dag_saturate_weights=saturate)
        for k in range(22):
            m.insert(*synth.cfg5_frame_torch(k))
        m.close_submap()
    m.sync()
    t0 = time.perf_counter()
    m.finalize()
    t = time.perf_counter() - t0
    st = m.dag_stats()
    levels = [m.dag_level(l) for l in range(21)]
    roots = [m.dag_roots(s) for s in range(m.closed_submaps)]
    m.close()
    return levels, roots, st, t


ok = True
for kind in ("empty", "cfg2", "cfg3", "cfg5"):
    for saturate in (False, True):
        dl, dr, ds, dt = build(kind, False, saturate)
        hl, hr, hs, ht = build(kind, True, saturate)
        same = dr == hr and all(np.array_equal(a, b) for a, b in zip(dl, hl)) and list(ds["uniques"]) == list(hs["uniques"]) and list(ds["dupes"]) == list(hs["dupes"])
        ok = ok and same
        bad = [l for l in range(21) if not np.array_equal(dl[l], hl[l])]
        print(f"{kind:6s} saturate={int(saturate)} {'SAME' if same else 'DIFFERENT levels ' + str(bad) + ' roots ' + str(dr) + ' vs ' + str(hr)}  device fold {dt * 1e3:.2f} ms (device {ds['ms_device']:.2f}), host fold {ht * 1e3:.2f} ms "
              f"(device {hs['ms_device']:.2f} host {hs['ms_host']:.2f}); cluster words {len(dl[20])}, level-19 words {len(dl[19])}", flush=True)
        if not same:
            for l in bad[:3]:
                n = min(len(dl[l]), len(hl[l])); d = np.flatnonzero(dl[l][:n] != hl[l][:n])
                print(f"   level {l}: sizes {len(dl[l])} vs {len(hl[l])}, first difference at {d[:5]}, uniques {ds['uniques'][l]} vs {hs['uniques'][l]}, dupes {ds['dupes'][l]} vs {hs['dupes'][l]}")
sys.exit(0 if ok else 1)

#!/usr/bin/env python3
"""Generates the small golden vectors under tests/golden/ (SURVEY.md section 8(c) G1-G6, and G8 for the finalize row 8(f1)).

PROVENANCE.  The reference itself cannot be built or run in this image (it needs glm, gtl, libmorton and fmt, none of
which is installed, and upstream ships no fixtures), so these vectors are produced by the oracle -- the CPU
restatement in oracle/chad_oracle.cpp -- AFTER that oracle reproduced the scalar summaries of a real reference run
recorded in SURVEY.md section 8(c) G7 (tests/test_oracle_survey_pins.py).  They freeze the oracle's behaviour so that
later edits to the oracle or the generators cannot drift unnoticed, and they give the GPU parity tests fixed inputs
with fixed expected outputs.  Run from the repo root:  python tests/golden/make_golden.py [names...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from chad_tsdf_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def g1_morton():
    rng = np.random.default_rng(101)
    pts = np.concatenate([
        (rng.standard_normal((3000, 3)) * 25.0).astype(np.float32),
        np.array([[0, 0, 0], [-0.0, 0.05, -0.05], [52428.75, -52428.8, 52428.7], [-52428.8, 52428.75, 0.0]], np.float32),  # +-(2^20 - 1) voxels at 0.05 m
        (np.arange(-200, 200, dtype=np.float32)[:, None] * np.float32(0.05)).repeat(3, axis=1),                             # exact lattice
    ])
    return dict(points=pts, res=np.float32(0.05), codes=po.morton_codes(pts, 0.05))


def g3_normals():
    pts, pos = synth.cfg2_frame(6000, seed=31)
    order, codes, nrm, starts = po.frame_normals(pts, pos, 0.05)
    return dict(points=pts, position=pos, res=np.float32(0.05), order=order, codes=codes, normals=nrm, group_start=starts)


def g4_rays():
    rng = np.random.default_rng(404)
    pos = np.array([0.3, -0.2, 0.1], np.float32)
    pts = np.concatenate([
        (rng.standard_normal((1500, 3)) * 4.0).astype(np.float32),
        np.array([[5, -0.2, 0.1], [0.3, 4, 0.1], [0.3, -0.2, -3], [2.3, 1.8, 0.1], [1.3, 0.8, 1.1]], np.float32),  # axis-aligned and diagonal
        (np.round(rng.standard_normal((300, 3)) * 3.0 / 0.05) * 0.05).astype(np.float32),                         # starts on voxel boundaries
    ])
    out = dict(points=pts, position=pos)
    for tag, res, trunc in (("a", 0.05, 0.10), ("b", 0.02, 0.06)):
        vox, cnt = po.walk_rays(pts, pos, res, trunc, 16)
        assert cnt.max() <= 16
        out[f"voxels_{tag}"] = vox.astype(np.int16) if np.abs(vox).max() < 32000 else vox
        out[f"counts_{tag}"] = cnt.astype(np.uint8)
        out[f"params_{tag}"] = np.array([res, trunc], np.float32)
    return out


def g5_leaves():
    out = {}
    for tag, (pts, pos), res, trunc in (("cfg1", (synth.plane_scan(100, 50), np.zeros(3, np.float32)), 0.10, 0.20),
                                        ("cfg2", synth.cfg2_frame(8000, seed=52), 0.05, 0.10)):
        m = po.OracleMap(res, trunc)
        m.insert(pts, pos)
        mc, sd, w = m.export_leaves()
        out.update({f"{tag}_points": pts, f"{tag}_position": pos, f"{tag}_params": np.array([res, trunc], np.float32),
                    f"{tag}_mc": mc, f"{tag}_sd": sd, f"{tag}_w": w.astype(np.uint16)})
    return out


def g6_sequence():
    """12 frames of 1500 points, poses 0.5 m apart along x: the 12th pose is 5.5 m from the first -> one roll-over."""
    m = po.OracleMap(0.05, 0.10)
    frames, poses = [], []
    for k in range(12):
        pts, _ = synth.cfg3_frame(10 * k, 1500)
        pos = np.array([-12.0 + 0.5 * k, 0.1, 0.0], np.float32)
        m.insert(pts, pos)
        frames.append(pts)
        poses.append(pos)
    assert m.closed_submaps == 1
    c_mc, c_sd, c_w = m.export_leaves(0)
    a_mc, a_sd, a_w = m.export_leaves(-1)
    return dict(points=np.stack(frames), positions=np.stack(poses), closed_mc=c_mc, closed_sd=c_sd, closed_w=c_w.astype(np.uint16),
                active_mc=a_mc, active_sd=a_sd, active_w=a_w.astype(np.uint16))


def g8_finalize():
    """Submap::finalize (submap.hpp:10-106): two submaps given as leaf lists, folded one after the other into one DAG.
    Expected: the 21 level arrays after each submap, the uniques/dupes counters and the roots."""
    m = po.OracleMap(0.05, 0.10)
    out = {}
    for k, seed in enumerate((81, 82)):
        src = po.OracleMap(0.05, 0.10)
        pts, pos = synth.cfg2_frame(2500, seed=seed)
        src.insert(pts, pos)
        mc, sd, w = src.export_leaves()
        m.close_submap_from_leaves(mc, sd, w)
        out.update({f"s{k}_mc": mc, f"s{k}_sd": sd, f"s{k}_w": w.astype(np.uint16), f"s{k}_roots": np.array(m.dag_roots(k), np.uint32)})
        for l in range(21):
            out[f"s{k}_level{l}"] = m.dag_level(l)
        out[f"s{k}_counts"] = np.array([m.dag_level_counts(l) for l in range(21)], np.uint32)
    return out


if __name__ == "__main__":
    every = (("g1_morton", g1_morton), ("g3_normals", g3_normals), ("g4_rays", g4_rays), ("g5_leaves", g5_leaves), ("g6_sequence", g6_sequence),
             ("g8_finalize", g8_finalize))
    for name, fn in every:
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **fn())
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")

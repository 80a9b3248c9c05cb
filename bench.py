#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config: inserted points/sec on the 1 M-point room scan
(cfg2: voxel 0.05 m, trunc 0.10 m), one process per GPU.

A step = chad::TSDFMap::insert() of one 1 M-point synthetic scan into an EMPTY map on this rank's GPU
(map.reset() + map.insert(), both inside the timed region), with the scan already resident in HBM.  insert() is
asynchronous, as in any streaming use: the library keeps up to three frames in flight (the map-independent stages
of the next frames run beside the update of the current one), and the timed region ends with a full synchronisation.  With N > 1
ranks every rank inserts its own scans (frames dealt round-robin: weak scaling) and the ranks' submaps are merged
once at the end of the timed region with the RCCL exchange of chad_tsdf_amd/sharded.py (all_gather of block keys
+ all_reduce of the per-block partial sums), which is the path's only collective.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      the TSDF-update kernel against the HBM roofline: algorithmic bytes (24 B/point + 16 B/distinct voxel,
                SURVEY.md section 8(d)) / its mean duration, measured with HIP events on the map's stream inside the
                timed region;
  cpu_baseline  the oracle (CPU restatement of the reference, single-threaded like the reference) on the same
                workload, timed on this host for a bounded number of frames.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (100 x 0.12 ms: the pipeline's fill and drain, ~0.2 ms, weigh 1-2 % instead of 8 % at 20)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=1_000_000, help="points per scan (BASELINE config: 1 M)")
    ap.add_argument("--frames", type=int, default=4, help="distinct synthetic scans cycled through per rank")
    ap.add_argument("--update-mode", type=int, default=0, help="0 auto, 1 scatter, 2 LDS-tiled")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from chad_tsdf_amd import TSDFMap, synth
    from chad_tsdf_amd.sharded import merge_blocks

    res, trunc = synth.CFG2["res"], synth.CFG2["trunc"]
    # scans of the cfg2 room from slightly different sensor poses / seeds; frame f of the job belongs to rank f % world
    scans = []
    for j in range(args.frames):
        f = j * world + rank
        sensor = (0.3 + 0.01 * f, -0.2, 0.1)
        pts, pos = synth.cfg2_frame(args.points, seed=synth.CFG2["seed"] + f, sensor=sensor)
        scans.append((torch.from_numpy(pts).cuda(), pos, pts))
    torch.cuda.synchronize()

    # the timed map brackets only the update kernel with HIP events (2 per insert; every hipEventRecord costs ~6 us of
    # stream time, so the full five-stage timing is taken from a separate, untimed pass below)
    m = TSDFMap(res, trunc, time_update=True, update_mode=args.update_mode, max_points=args.points)

    def step(i):
        d_pts, pos, _ = scans[i % len(scans)]
        m.reset()
        m.insert(d_pts, pos)

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    m.sync()
    if world > 1:  # warm the collective path once as well
        merge_blocks(*m.export_blocks(-1, ordered=False))
    m.stats()  # drop the warm-up's stage timings
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    t_enqueued = time.perf_counter() - t0  # host side only: all steps are enqueued, the device is still working
    if world > 1:
        merged_keys, merged_sum, merged_cnt = merge_blocks(*m.export_blocks(-1, ordered=False))
    m.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = m.stats()  # ms_oct_upd: mean duration of the update kernel over the timed steps (HIP events on the map's stream)
    # the reference's five stage timers: a separate untimed pass over the same scans with full stage timing (stage timing
    # makes the map run its frames one after the other on one stream, so the stage times do not overlap)
    ms_ = TSDFMap(res, trunc, timing=True, update_mode=args.update_mode, max_points=args.points)
    for i in range(min(8, args.steps) + 1):
        ms_.reset()
        ms_.insert(scans[i % len(scans)][0], scans[i % len(scans)][1])
        if i == 0:
            ms_.stats()  # drop the first (cold) insert
    stage = ms_.stats()
    ms_.close()
    # the update kernel on its own: one frame at a time (sync after every insert), timed like the live figure by the dispatch's
    # begin / end time stamps -- no event records around it, nothing else on the chip
    alone = TSDFMap(res, trunc, time_update=True, update_mode=args.update_mode, max_points=args.points)
    for i in range(min(8, args.steps) + 1):
        alone.reset()
        alone.insert(scans[i % len(scans)][0], scans[i % len(scans)][1])
        alone.sync()
        if i == 0:
            alone.stats()
    upd_alone_ms = float(alone.stats()["ms_oct_upd"])
    alone.close()
    mc, sd, w = m.export_leaves()  # of the last step's map: V for the roofline's algorithmic bytes
    n_voxels = int(len(mc))
    algo_bytes = 24 * args.points + 16 * n_voxels  # K7 = 24 B/point + 16 B/distinct voxel
    upd_ms = float(st["ms_oct_upd"])
    achieved = algo_bytes / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_update_kernel.json")  # written from a separate rocprofv3 --pmc run
    if os.path.exists(pmc_file):
        try:
            traffic = json.load(open(pmc_file)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "inserted points/sec",
        "value": world * args.steps * args.points / elapsed,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "cfg2: single 1M-point synthetic room scan (10x8x3 m box), voxel 0.05 m, trunc 0.10 m, inserted into an empty map each step",
            "points_per_frame": args.points,
            "frames_per_step_per_gpu": 1,
            "parallelism": f"frames round-robin over {world} GPU(s); one RCCL block-partial merge per timed region" if world > 1 else "1 GPU",
            "update_kernel": {0: "auto (tiled for this dense frame)", 1: "scatter", 2: "tiled"}[args.update_mode],
        },
        "host_enqueue_ms_per_step": round(t_enqueued / args.steps * 1e3, 4),
        "stage_ms": {k: round(float(stage[k]), 4) for k in ("ms_mc_calc", "ms_mc_sort", "ms_norm_est", "ms_oct_upd", "ms_total")},
        "frame": {"updates": int(st["updates"]), "voxels": n_voxels, "blocks": int(st["blocks"]), "sort_passes": int(st["sort_passes"]),
                  "hash_load": round(st["blocks"] / max(1, st["hash_capacity"]), 5)},
        "roofline": {
            "kernel": "k_update_tiled" if args.update_mode != 1 else "k_update_scatter",
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "algorithmic_bytes_per_launch": algo_bytes,
            "kernel_ms": round(upd_ms, 4),
            # the timed region keeps three frames in flight (the preparation of the next frames runs beside the update), so the
            # live duration above includes that contention; the same kernel with one frame at a time on the chip:
            "kernel_ms_serial": round(upd_alone_ms, 4),
            "frac_serial": round(algo_bytes / (upd_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if upd_alone_ms > 0 else None,
        },
    }

    if rank == 0 and world == 1:
        # the row behind the path, outside the timed region: the last step's map closed and folded
        # into the DAG of NodeLevels (Submap::finalize, submap.hpp:10-106)
        m.close_submap()
        m.sync()
        t1 = time.perf_counter()
        m.finalize()
        fin_ms = (time.perf_counter() - t1) * 1e3
        ds = m.dag_stats()
        out["finalize"] = {"ms": round(fin_ms, 3), "ms_device": round(ds["ms_device"], 3), "ms_host": round(ds["ms_host"], 3),
                           "blocks": int(st["blocks"]), "leaves": n_voxels, "unique_clusters": ds["uniques"][20],
                           "dag_words": int(sum(len(m.dag_level(l)) * (2 if l == 20 else 1) for l in range(21)))}

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from oracle import pyoracle  # CPU baseline leg only: the checker, timed -- never on the product path

        o = pyoracle.OracleMap(res, trunc)
        frames, t_cpu = 0, 0.0
        stage = np.zeros(5)
        while t_cpu < args.cpu_seconds and frames < 200:
            _, pos, pts = scans[frames % len(scans)]
            t1 = time.perf_counter()
            o.reset()
            o.insert(pts, pos)
            t_cpu += time.perf_counter() - t1
            s = o.stats()
            stage += np.array([s["ms_calc"], s["ms_sort"], s["ms_norm"], s["ms_upd"], s["ms_total"]])
            frames += 1
        cpu_model = ""
        try:
            cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        out["cpu_baseline"] = {
            "value": frames * args.points / t_cpu,
            "unit": "points/s",
            "cores": 1,
            "kind": "port",
            "sample": f"{frames} of the same 1M-point cfg2 scans, reset + insert each, {t_cpu:.1f} s on 1 of {os.cpu_count()} cores ({cpu_model}); "
                      "the reference is single-threaded (no threads anywhere; par_unseq commented out at morton.hpp:88)",
            "ms_per_frame": round(t_cpu / frames * 1e3, 2),
            "stage_ms": dict(zip(("mc_calc", "mc_sort", "norm_est", "oct_upd", "total"), (stage / frames).round(2).tolist())),
        }
        out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        o.reset()
        o.close_submap_from_leaves(mc, sd, w)  # the oracle's finalize of the same leaves, on one core
        out["finalize"]["cpu_ms"] = round(o.finalize_ms, 3)
        out["finalize"]["levels_identical"] = bool(all(np.array_equal(m.dag_level(l), o.dag_level(l)) for l in range(21)))

    if rank == 0:
        print(json.dumps(out))
    m.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

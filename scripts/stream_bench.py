#!/usr/bin/env python3
"""cfg3 / cfg4 of BASELINE.json: a stream of 200 k-point frames from a moving sensor (0.05 m per frame, roll-over every 101
frames), device-resident, inserted back to back -- on one GPU (cfg3), or dealt round-robin over N ranks with the block
partials merged at every submap close (cfg4, strong scaling):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/stream_bench.py --gpus N
(add --backend gloo to rehearse a few ranks on one GPU).  Prints steady-state throughput and the hash occupancy (the CPU path on the same frames:
tests/soak/stream_cpu_baseline.py).  Not the bench line (bench.py is): a measurement script for DESIGN.md."""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chad_tsdf_amd import TSDFMap, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=500)
ap.add_argument("--gpus", type=int, default=1, help="cfg4: the same stream dealt round-robin over this many ranks (launch with torch.distributed.run)")
ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) on a multi-GPU node; gloo to rehearse several ranks on one GPU")
args = ap.parse_args()
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank % torch.cuda.device_count())
assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
if world > 1:
    import torch.distributed as dist
    from chad_tsdf_amd.sharded import ShardedTSDFMap

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group(args.backend)
# a rank needs the points of its own frames only (frame k belongs to rank k % world); every rank sees every pose
poses = [np.asarray((-12.0 + 0.05 * k, 0.1, 0.0), dtype=np.float32) for k in range(args.frames)]
frames = [synth.cfg3_frame(k) if k % world == rank else (None, poses[k]) for k in range(args.frames)]
dev = [torch.from_numpy(p).cuda() if p is not None else None for p, _ in frames]
torch.cuda.synchronize()
if world > 1:
    m = ShardedTSDFMap(0.05, 0.10)
    local = m.local
    for k in range(min(2 * world, args.frames)):                         # warm the kernels ...
        m.insert(dev[k], poses[k])
    m.merge_and_close()                                                  # ... and the collectives
    local.reset(); local.sync()
    m = ShardedTSDFMap(0.05, 0.10, local=local)
    dist.barrier()
else:
    m = local = TSDFMap(0.05, 0.10)
    for (_, pos), d in list(zip(frames, dev))[:5]:
        m.insert(d, pos)
    m.sync(); m.reset(); m.sync()
t0 = time.perf_counter()
for pos, d in zip(poses, dev):
    m.insert(d, pos)
t_host = time.perf_counter() - t0
if world > 1:
    m.merge_and_close()   # the last submap's exchange belongs to the job
local.sync()
if world > 1:
    dist.barrier()
t = time.perf_counter() - t0
if world > 1:
    tt = torch.tensor([t, t_host], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t, t_host = float(tt[0]), float(tt[1])
st = local.stats()
out = {"workload": "cfg%d: %d frames x 200k points, voxel 0.05 m, trunc 0.10 m, sensor moving 0.05 m/frame%s" % (
           4 if world > 1 else 3, args.frames, ", frames round-robin over %d ranks (%s), block partials merged at every submap close" % (world, args.backend) if world > 1 else ""),
       "n_gpus": world, "scaling": "strong",
       "ms_per_frame": t / args.frames * 1e3, "points_per_s": args.frames * 200_000 / t, "host_enqueue_ms_per_frame": t_host / args.frames * 1e3,
       "closed_submaps": st["closed_submaps"], "live_blocks": st["blocks"], "block_capacity": st["block_capacity"],
       "hash_load": st["blocks"] / st["hash_capacity"], "overflow": st["overflow"]}
if rank != 0:
    if world > 1:
        dist.destroy_process_group()
    sys.exit(0)
print(json.dumps(out))
if world > 1:
    dist.destroy_process_group()

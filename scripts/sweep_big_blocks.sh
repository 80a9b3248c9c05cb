#!/bin/bash
# k_frame_normals: which point blocks get a whole workgroup (size threshold) and how many workgroups serve them
for bb in 128 256 384 512 1024; do for bg in 128 256 512 768; do
  echo "big_block=$bb big_groups=$bg $(CHAD_DEBUG_BIG_BLOCK=$bb CHAD_DEBUG_BIG_GROUPS=$bg python bench.py --steps 30 --warmup 3 --cpu-seconds 0 --no-extras 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); k={x["kernel"]:x["ms"] for x in d["roofline"]["kernels"]}; print("normals", k["k_frame_normals"], "gather", k["k_frame_gather"], "ms_per_step", round(d["ms_per_step"],4))')"
done; done | tee gpurun_out/sweep_big_blocks.log

#!/bin/bash
# normals kernel (alone) against the size above which a block gets a whole workgroup and the number of such workgroups
for bb in 192 256 384; do for g in 384 512 640; do
  echo "big_block=$bb big_groups=$g $(CHAD_DEBUG_BIG_BLOCK=$bb CHAD_DEBUG_BIG_GROUPS=$g python bench.py --steps 20 --cpu-seconds 0 --no-extras 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("normals alone", d["stage_ms"]["ms_norm_est"], "ms/frame", round(d["ms_per_frame"],4))')"
done; done 2>&1 | tee gpurun_out/sweep_big_blocks.log

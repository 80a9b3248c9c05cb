#!/bin/bash
# How much of k_frame_normals is the few LARGE point blocks: blocks above a size are skipped (timing only, results are wrong).
for l in 0 2048 1024 512 256 128; do
  echo "maxlen=$l norm_ms $(CHAD_DEBUG_ABLATE=$((l << 16)) python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-extras | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["stage_ms"]["ms_norm_est"])')"
done 2>&1 | grep -v amdgpu | tee gpurun_out/ablate_normals_len.log

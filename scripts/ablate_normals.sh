#!/bin/bash
# Where k_frame_normals spends its time: CHAD_DEBUG_ABLATE bits switch phases off (timing only, results are wrong):
#   8 all fits (moments and planes), 16 the planes only, 32 output pass, 64 greedy walk (no neighbourhoods at all)
for a in 0 16 8 32 40 64 104; do
  echo "ablate=$a norm_ms $(CHAD_DEBUG_ABLATE=$a python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-extras | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["stage_ms"]["ms_norm_est"])')"
done 2>&1 | grep -v amdgpu | tee gpurun_out/ablate_normals.log

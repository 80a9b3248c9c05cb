#!/bin/bash
# the bare re-ordering of k_frame_normals (fits, output and walk off: CHAD_DEBUG_ABLATE bits 8 | 32 | 64) against the largest block worked on
for l in 0 512 256 128 64; do
  echo "bare maxlen=$l norm_ms $(CHAD_DEBUG_ABLATE=$(((l << 16) | 104)) python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-extras | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["stage_ms"]["ms_norm_est"])')"
done 2>&1 | grep -v amdgpu | tee gpurun_out/ablate_normals_bare.log

#!/bin/bash
for a in 0 64 8 16 24 32 56 120; do
  echo "ablate=$a norm_ms $(CHAD_DEBUG_ABLATE=$a python bench.py --steps 10 --warmup 3 --cpu-seconds 0 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["stage_ms"]["ms_norm_est"])')"
done 2>&1 | grep -v amdgpu | tee gpurun_out/ablate_normals.log

#!/bin/bash
mkdir -p gpurun_out
for a in 0 1 16; do
  echo "== ablate=$a"; CHAD_DEBUG_STAMPS=1 CHAD_DEBUG_ABLATE=$a python bench.py --steps 5 --warmup 2 --cpu-seconds 0 2>&1 | grep -E "stamps|metric" | cut -c1-400
done 2>&1 | tee gpurun_out/stamps.log

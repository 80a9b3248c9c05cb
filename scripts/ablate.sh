#!/bin/bash
# timing experiments on the update kernel: CHAD_DEBUG_ABLATE bits (diagnostics only; results are wrong by design)
mkdir -p gpurun_out
for a in 0 1 16 8 2 4 6 18 24; do
  echo "ablate=$a $(CHAD_DEBUG_ABLATE=$a python bench.py --steps 20 --warmup 3 --cpu-seconds 0 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["roofline"]["kernel_ms"], d["ms_per_step"], d["stage_ms"])')"
done 2>&1 | tee gpurun_out/ablate.log

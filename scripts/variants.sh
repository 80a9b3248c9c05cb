#!/bin/bash
mkdir -p gpurun_out
for v in 0 1; do
 echo "variant $v: $(CHAD_DEBUG_TILE_VARIANT=$v python bench.py --cpu-seconds 0 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), d["roofline"]["kernel_ms"], d["stage_ms"])')"
done 2>&1 | tee gpurun_out/variants.log

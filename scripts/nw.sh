#!/bin/bash
for n in 24 16 12 8; do
 echo "normal waves/CU $n: $(CHAD_DEBUG_NORMAL_WAVES=$n python bench.py --cpu-seconds 0 --steps 40 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), d["roofline"]["kernel_ms"], d["stage_ms"]["ms_norm_est"])')"
done 2>&1 | grep -v amdgpu | tee gpurun_out/nw.log

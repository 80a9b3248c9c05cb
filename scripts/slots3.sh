#!/bin/bash
for q in 4 8; do for n in 3 4 5; do
 echo "no-priority hwq $q slots $n: $(CHAD_DEBUG_NO_PRIORITY=1 GPU_MAX_HW_QUEUES=$q CHAD_SLOTS=$n python bench.py --cpu-seconds 0 --steps 40 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), d["roofline"]["kernel_ms"])')"
done; done 2>&1 | grep -v amdgpu | tee gpurun_out/slots3.log

#!/bin/bash
for n in 1 2 3 4 6; do
 echo "slots $n: $(CHAD_SLOTS=$n python bench.py --cpu-seconds 0 --steps 40 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), d["roofline"]["kernel_ms"])')"
done 2>&1 | grep -v amdgpu | tee gpurun_out/slots.log
echo "200k points:"
for n in 1 3 6; do
 echo "slots $n: $(CHAD_SLOTS=$n python bench.py --cpu-seconds 0 --steps 40 --points 200000 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), d["roofline"]["kernel_ms"])')"
done 2>&1 | grep -v amdgpu | tee -a gpurun_out/slots.log

#!/bin/bash
for n in 2 3 4; do for pr in 0 1; do
 if [ $pr = 1 ]; then export CHAD_DEBUG_NO_PRIORITY=1; else unset CHAD_DEBUG_NO_PRIORITY; fi
 echo "slots $n nopriority $pr: $(CHAD_SLOTS=$n python bench.py --cpu-seconds 0 --steps 40 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), d["roofline"]["kernel_ms"])')"
done; done 2>&1 | grep -v amdgpu

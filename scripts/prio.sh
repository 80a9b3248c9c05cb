#!/bin/bash
for st in 20 40 100; do for pr in 0 1; do
 if [ $pr = 1 ]; then export CHAD_DEBUG_NO_PRIORITY=1; else unset CHAD_DEBUG_NO_PRIORITY; fi
 for i in 1 2; do echo "steps $st nopriority $pr: $(python bench.py --cpu-seconds 0 --steps $st | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), d["roofline"]["kernel_ms"], d["host_enqueue_ms_per_step"])')"; done
done; done 2>&1 | grep -v amdgpu

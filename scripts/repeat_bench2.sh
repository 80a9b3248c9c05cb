#!/bin/bash
for i in 1 2 3; do python bench.py --cpu-seconds 3 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('cpu3', round(d['ms_per_step'],4), d['roofline']['kernel_ms'], d['host_enqueue_ms_per_step'])"; done 2>&1 | grep -v amdgpu
for i in 1 2 3; do python bench.py --cpu-seconds 0 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('cpu0', round(d['ms_per_step'],4), d['roofline']['kernel_ms'], d['host_enqueue_ms_per_step'])"; done 2>&1 | grep -v amdgpu

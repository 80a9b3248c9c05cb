#!/bin/bash
# A/B helper: N runs of the contract line (400 steps), prints points/s (G) and the serial normals time
n=${1:-5}; shift
for i in $(seq $n); do "$@" python bench.py --steps 400 --no-extras --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k={x['kernel']:x['ms'] for x in d['roofline']['kernels']}; print(round(d['value']/1e9,3), round(d['ms_per_step'],4), 'normals', k['k_frame_normals'], 'sum', d['roofline']['kernel_ms_sum_serial'])"; done

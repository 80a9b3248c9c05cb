#!/bin/bash
# A/B helper: N runs of the contract line (100 steps of 4 frames), prints points/s (G), ms per frame, the serial normals time and kernel sum
n=${1:-5}; shift
for i in $(seq $n); do "$@" python bench.py --steps 100 --no-extras --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k={x['kernel']:x['ms'] for x in d['roofline']['kernels']}; print(round(d['value']/1e9,3), round(d['ms_per_frame'],4), 'normals', k['k_frame_normals'], 'sum', d['roofline']['kernel_ms_sum_serial'])"; done

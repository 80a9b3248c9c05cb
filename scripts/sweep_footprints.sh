#!/bin/bash
# the two big kernels with fewer resident waves (room on every CU for the other frames' kernels): update workgroups per CU x normals waves per CU
for u in 5 4 3; do for nw in 20 16 12; do
  echo "update wg/cu=$u normals waves/cu=$nw $(CHAD_DEBUG_WAVES_PER_CU=$u CHAD_DEBUG_NORMAL_WAVES=$nw python bench.py --cpu-seconds 0 --no-extras 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("ms/frame", round(d["ms_per_frame"],4), "upd live", d["roofline"]["kernel_ms"], d["stage_ms"])')"
done; done 2>&1 | tee gpurun_out/sweep_footprints.log

#!/bin/bash
# Pipelined step time against the chip share the two big kernels take: update workgroups per CU x normals waves per CU
for u in 3 4 5; do for nw in 12 16 24; do
  echo "update_wg_per_cu=$u normal_waves_per_cu=$nw $(CHAD_DEBUG_WAVES_PER_CU=$u CHAD_DEBUG_NORMAL_WAVES=$nw python bench.py --steps 100 --warmup 3 --cpu-seconds 0 --no-extras 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); k={x["kernel"]:x["ms"] for x in d["roofline"]["kernels"]}; print("ms_per_step", round(d["ms_per_step"],4), "upd_live", d["roofline"]["kernel_ms"], "upd_serial", k["k_update"], "normals_serial", k["k_frame_normals"])')"
done; done | tee gpurun_out/sweep_footprints.log

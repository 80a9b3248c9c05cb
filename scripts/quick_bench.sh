#!/bin/bash
# the contract's line only (no extra legs), for a few settings of the frames in flight
mkdir -p gpurun_out
for s in ${1:-3}; do
  echo "slots=$s $(CHAD_SLOTS=$s python bench.py --cpu-seconds 0 --no-extras 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("ms/frame", round(d["ms_per_frame"],4), "Gpts/s", round(d["value"]/1e9,3), "upd live", r["kernel_ms"], "serial", r.get("kernel_ms_serial"), d["stage_ms"])')"
done 2>&1 | tee gpurun_out/quick_bench.log

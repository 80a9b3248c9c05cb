#!/bin/bash
# frames in flight (CHAD_SLOTS) at half and full frame size: is it the footprint -- three 1 M-point frames = 230 MB of buffers, the Infinity Cache holds 256 MB -- that makes four slots slower?  (No: the same at 500 k points.)
for n in 500000 1000000; do for s in 2 3 4; do
  echo "points=$n slots=$s $(CHAD_SLOTS=$s python bench.py --cpu-seconds 0 --no-extras --points $n 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("ms/frame", round(d["ms_per_frame"],4), "Gpts/s", round(d["value"]/1e9,3), "upd live", d["roofline"]["kernel_ms"], "serial sum", d["stage_ms"]["ms_total"])')"
done; done 2>&1 | tee gpurun_out/slots_footprint.log

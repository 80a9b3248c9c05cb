#!/bin/bash
# in-kernel phase stamps of the update kernel (diagnostic instantiation): frames in flight, then one frame at a time
mkdir -p gpurun_out
for serial in 0 1; do
  echo "== CHAD_DEBUG_SERIAL=$serial"; CHAD_DEBUG_SERIAL=$serial CHAD_DEBUG_STAMPS=1 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-extras 2>&1 | grep -E "stamps|metric" | cut -c1-400
done 2>&1 | tee gpurun_out/stamps.log

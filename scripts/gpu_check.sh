#!/bin/bash
# one GPU call: parity tests, bench line, stamps of the update kernel
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/tests.log
timeout -k 10 200 python bench.py --cpu-seconds 0 > gpurun_out/bench.log 2>&1 && python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
print("ms/step",round(d["ms_per_step"],4),"Gpts/s",round(d["value"]/1e9,3),"upd_ms",d["roofline"]["kernel_ms"],"frac",d["roofline"]["frac"],d["stage_ms"],d["frame"])
PY
CHAD_DEBUG_STAMPS=1 timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 2>&1 | grep stamps | tail -2

#!/bin/bash
# one GPU call: parity tests, then (only if they pass) the bench line and the in-kernel stamps of the update kernel
mkdir -p gpurun_out
tag=${1:-check}
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/${tag}_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --cpu-seconds 0 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/${tag}_bench.json') if l.startswith('{')][-1])
r=d["roofline"]
print("ms/frame",round(d["ms_per_frame"],4),"Gpts/s",round(d["value"]/1e9,3),"upd live ms",r["kernel_ms"],"frac",r["frac"],"serial",r["kernel_ms_serial"],r["frac_serial"])
print("kernels", [(k["kernel"],k["ms"]) for k in r["kernels"]], "sum", r["kernel_ms_sum_serial"], "large", r["large_frame"]["kernel_ms"], r["large_frame"]["frac"])
print("cfg3", d["cfg3"]["ms_per_frame"], "cfg5", d["cfg5"]["ms_per_frame"], d["cfg5"]["update_kernel_ms"], "pcie", d["pcie_inclusive"]["ms_per_frame"], "finalize", d["finalize"]["ms"])
PY
CHAD_DEBUG_SERIAL=1 CHAD_DEBUG_STAMPS=1 timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-extras 2>&1 | grep stamps | tail -4

#!/bin/bash
# parity tests of the frame path + the contract's line with the per-kernel and large-frame legs (no streams, no CPU leg)
tag=${1:-quick}
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_properties.py -x -q > "$out/tests.log" 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 "$out/tests.log"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --cpu-seconds 0 --stream-frames 0 --outdoor-frames 0 > "$out/bench.json" 2> "$out/bench.err"; rc=$?
python3 - "$out/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("cfg2 ms/frame", round(d["ms_per_frame"], 4), "Gpts/s", round(d["value"] / 1e9, 3), "upd live", r["kernel_ms"], "frac", r["frac"], "serial", r.get("kernel_ms_serial"), r.get("frac_serial"), "large", r.get("large_frame", {}).get("kernel_ms"), r.get("large_frame", {}).get("frac"))
print("kernel sum serial", r.get("kernel_ms_sum_serial"), [(k["kernel"], k["ms"]) for k in r.get("kernels", [])])
print("pcie", d.get("pcie_inclusive", {}).get("ms_per_frame"))
PY
exit $rc

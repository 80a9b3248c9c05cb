#!/bin/bash
# One GPU call of a round's build: the driver's bench command FIRST (fresh box, as the driver runs it), then the GPU tests.
#   gpurun --timeout 900 -- 'bash scripts/round_check.sh r4a'
tag=${1:-check}
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; rc=$?
echo "bench rc=$rc"
python3 - "$out/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line:", e); sys.exit(0)
r = d["roofline"]
print("cfg2 ms/frame", round(d["ms_per_frame"], 4), "Gpts/s", round(d["value"] / 1e9, 3), "upd frac", r["frac"], "serial", r.get("frac_serial"), "large", r.get("large_frame", {}).get("frac"))
print("kernel sum serial", r.get("kernel_ms_sum_serial"), [(k["kernel"], k["ms"]) for k in r.get("kernels", [])])
for leg in ("cfg3", "cfg5"):
    if leg in d:
        print(leg, json.dumps({k: v for k, v in d[leg].items() if k != "workload"}))
PY
[ $rc -eq 0 ] || exit $rc
timeout -k 10 ${2:-600} python3 -m pytest tests -m gpu -x -q > "$out/tests.log" 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 "$out/tests.log"
exit $rc

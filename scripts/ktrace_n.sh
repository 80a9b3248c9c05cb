#!/bin/bash
# per-kernel durations at another frame size: bash scripts/ktrace_n.sh <points>
n=${1:-200000}
out=gpurun_out/ktrace_$n
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --points $n > "$out/bench.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    print(f'{r["Name"].split("(")[0][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us min {float(r["MinNs"])/1e3:8.2f} max {float(r["MaxNs"])/1e3:8.2f}')
PY
grep -o '"ms_per_step": [0-9.]*' "$out/bench.log"

#!/bin/bash
out=gpurun_out/ktrace_outdoor
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 scripts/outdoor_bench.py --frames 40 > "$out/bench.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    print(f'{r["Name"].split("(")[0][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Percentage"]:>6s}%')
PY
tail -1 "$out/bench.log"

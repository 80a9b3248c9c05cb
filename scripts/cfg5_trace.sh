#!/bin/bash
# kernel-trace statistics of BASELINE config 5's trajectory (110 frames = 10 roll-overs): with the closed submaps folded into the DAG on
# the device and released (CHAD_FLAG_DAG_ONLY), and with their dense archives kept.  Run on the GPU box: bash scripts/cfg5_trace.sh <tag>
set -e
tag=${1:-r04}
out=gpurun_out/cfg5_$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
python3 scripts/cfg5_dagonly.py --frames 110 > "$out/dagonly.json" 2> "$out/dagonly.err"
python3 scripts/cfg5_dagonly.py --frames 110 --keep > "$out/keep.json" 2> "$out/keep.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_dagonly" -- python3 scripts/cfg5_dagonly.py --frames 110 > "$out/trace_dagonly.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_keep" -- python3 scripts/cfg5_dagonly.py --frames 110 --keep > "$out/trace_keep.log" 2>&1
for m in dagonly keep; do
  f=$(find "$out/trace_$m" -name '*kernel_stats.csv' | head -1)
  cp "$f" "$out/${m}_kernel_stats.csv"
done
cat "$out/dagonly.json" "$out/keep.json"

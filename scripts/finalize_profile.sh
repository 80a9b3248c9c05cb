#!/bin/bash
# kernel-trace stats of Submap::finalize (tests/soak/finalize_bench.py): the cluster kernel, the key sort and the copies
out=gpurun_out/finalize_prof
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 tests/soak/finalize_bench.py > "$out/bench.log" 2>&1
cat "$out/bench.log" | grep -v amdgpu.ids
python3 - "$out" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows:
    if any(k in r["Name"] for k in ("k_leaf_clusters", "k_copy_u64", "k_compact_bricks")):
        print(f'{r["Name"].split("(")[0][:50]:50s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.2f} us min {float(r["MinNs"])/1e3:8.2f} max {float(r["MaxNs"])/1e3:8.2f}')
t = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
for r in csv.DictReader(open(t)):
    if "k_leaf_clusters" in r["Kernel_Name"]:
        print("k_leaf_clusters launch: grid", r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), "duration us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
cp "$f" /dev/null 2>/dev/null; cp $(ls $out/*/*kernel_stats.csv | tail -1) gpurun_out/finalize_kernel_stats.csv

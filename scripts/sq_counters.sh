#!/bin/bash
# SQ counters of one kernel (default: the update kernel), mean per launch.  Run ON THE GPU BOX: gpurun -- 'bash scripts/sq_counters.sh [kernel substring]'
kern=${1:-k_update_tiled}
out=gpurun_out/sq
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$out"; mkdir -p "$out"
rocprofv3 -L > "$out/counters_list.txt" 2>&1
pass() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$out/$1" -- python3 bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-extras > "$out/$1.log" 2>&1 || echo "pass $1 failed"; }
pass a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
pass b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
pass c "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
pass d "GRBM_GUI_ACTIVE GRBM_COUNT"
python3 - "$out" "$kern" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
for name in kern.split(","):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if name in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", name)
    for k in sorted(acc):
        v = acc[k]
        print(f"{k:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY

#!/bin/bash
# the fold's grid size (workgroups of the multi-step kernels) against fold time: standalone (dag_device_check) and beside the frames (cfg5)
out=gpurun_out/${1:-fold_ab}; mkdir -p $out
for g in 64 128 256 512; do
  echo "== grid $g" >> $out/ab.log
  CHAD_DEBUG_FOLD_GRID=$g timeout -k 10 200 python3 tests/soak/dag_device_check.py 2>&1 | grep "fold" | awk '{print $1,$2,$3,$6,$7}' >> $out/ab.log || exit 1
  CHAD_DEBUG_FOLD_GRID=$g timeout -k 10 120 python3 scripts/cfg5_dagonly.py --frames 440 2>&1 | grep ms_per_frame >> $out/ab.log || exit 1
done
cat $out/ab.log

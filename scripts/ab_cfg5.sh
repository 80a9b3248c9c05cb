#!/bin/bash
# cfg5 under CHAD_FLAG_DAG_ONLY (440 frames) with this tree's library against another build's (a tree under scratch/), alternating on one box
other=${1:-scratch/prev}
for i in 1 2 3; do
  echo -n "here  "; timeout -k 10 120 python3 scripts/cfg5_dagonly.py --frames 440 2>&1 | grep -o '"ms_per_frame": [0-9.]*' || exit 1
  echo -n "other "; CHAD_AB_ROOT=$PWD/$other timeout -k 10 120 python3 scripts/cfg5_dagonly.py --frames 440 2>&1 | grep -o '"ms_per_frame": [0-9.]*' || exit 1
done

#!/bin/bash
# per-kernel times of cfg5 under CHAD_FLAG_DAG_ONLY (110 frames, rocprofv3 kernel trace) of this tree's library and of another build's (a tree under scratch/)
other=${1:-scratch/prev}; pattern=${2:-k_sorted_clusters}
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
for side in here other; do
  out=gpurun_out/abk_$side; rm -rf $out
  if [ $side = other ]; then export CHAD_AB_ROOT=$PWD/$other; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 scripts/cfg5_dagonly.py --frames 110 > /dev/null 2>&1 || exit 1
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  echo "== $side"; python3 - "$f" "$pattern" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(p in r["Name"] for p in sys.argv[2].split(",")): print(f'{r["Name"].split("(")[0][:60]:60s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
done
bash scripts/ab_cfg5.sh $other

#!/bin/bash
# this tree with an environment setting against another build's tree, alternating on one box: scripts/ab_env.sh scratch/prev CHAD_DEBUG_NORMAL_WAVES=24
other=${1:-scratch/prev}; setting=${2:-X=1}
line() { python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];k={e['kernel']:e['ms'] for e in r.get('kernels',[])};print('$1', round(d['value']/1e9,3),'G pts/s', round(d['ms_per_frame'],4),'ms/frame; alone: bounds',k.get('k_frame_bounds'),'keys',k.get('k_frame_keys'),'sort',k.get('k_radix_onesweep[0]'),k.get('k_radix_onesweep[1]'),'gather',k.get('k_frame_gather'),'normals',k.get('k_frame_normals'),'update',r['kernel_ms_serial'],'sum',r.get('kernel_ms_sum_serial'))"; }
for i in 1 2 3; do
  env $setting python3 bench.py --steps 100 --no-extras --cpu-seconds 0 2>/dev/null | line "this" || exit 1
  (cd $other && python3 bench.py --steps 100 --no-extras --cpu-seconds 0 2>/dev/null | line "prev") || exit 1
done

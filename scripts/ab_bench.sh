#!/bin/bash
# the headline bench line (pipelined value, the update kernel's time in flight and alone) of this tree against another build's tree, alternating on one box
other=${1:-scratch/prev}
for i in 1 2 3; do
  for root in . $other; do
    (cd $root && python3 bench.py --steps 100 --no-extras --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('$root', round(d['value']/1e9,3),'G pts/s', d['ms_per_frame'].__round__(4),'ms/frame; update in flight',r['kernel_ms'],'alone',r['kernel_ms_serial'])") || exit 1
  done
done

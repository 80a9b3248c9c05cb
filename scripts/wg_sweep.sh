#!/bin/bash
# update-kernel workgroups per CU (CHAD_DEBUG_WAVES_PER_CU): serial and pipelined
for serial in 1 0; do for w in 2 3 4 5; do
  if [ $serial = 1 ]; then export CHAD_DEBUG_SERIAL=1; else unset CHAD_DEBUG_SERIAL; fi
  CHAD_DEBUG_WAVES_PER_CU=$w python bench.py --cpu-seconds 0 --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('serial $serial wg/cu $w ms/step', round(d['ms_per_step'],4), 'upd live', d['roofline']['kernel_ms'], 'stage upd', d['stage_ms']['ms_oct_upd'], 'total', d['stage_ms']['ms_total'])"
done; done

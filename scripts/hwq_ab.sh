#!/bin/bash
# the number of hardware queues the HIP runtime spreads its streams over (GPU_MAX_HW_QUEUES, default 4) against the headline step, the cfg3
# stream and cfg5 with and without CHAD_FLAG_DAG_ONLY: a map keeps three frames in flight on a stream each, beside its own stream, the
# caller's and -- under DAG_ONLY -- two more
for q in 4 8 12 4 8; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python3 scripts/ab_headline.py child "$PWD" 2>&1 | tail -1 || exit 1
  GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python3 scripts/cfg5_dagonly.py --frames 440 2>&1 | grep -o '"ms_per_frame": [0-9.]*' || exit 1
done

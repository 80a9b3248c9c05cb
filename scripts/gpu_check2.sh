#!/bin/bash
# parity tests twice: normal, and with the guarded sort tail forced to do all passes but the first
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/tests.log
CHAD_DEBUG_SORT_PASSES=1 timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/tests_tail.log 2>&1; echo "tests (sort tail forced) rc=$?"; tail -3 gpurun_out/tests_tail.log
for n in 1000000 200000; do
python bench.py --cpu-seconds 0 --steps 40 --points $n | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('points', d['config']['points_per_frame'], 'ms/step', round(d['ms_per_step'],4), 'host', d['host_enqueue_ms_per_step'], 'upd', d['roofline']['kernel_ms'], d['stage_ms'])"
done

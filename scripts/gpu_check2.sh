#!/bin/bash
# parity tests twice: normal, and with the sort's hint forced to one launch (the last launch -- or, for grids too large to be resident, the catch-all -- does all passes but the first)
mkdir -p gpurun_out
tag=${1:-check2}
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/${tag}_tests.log
[ $rc -eq 0 ] || exit $rc
CHAD_DEBUG_SORT_PASSES=1 timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests_tail.log 2>&1; rc=$?; echo "tests (sort tail forced) rc=$rc"; tail -3 gpurun_out/${tag}_tests_tail.log
[ $rc -eq 0 ] || exit $rc
bash scripts/quick_bench.sh "3"

#!/bin/bash
for f in 0 1 2 3 4; do echo "flags=$f $(CHAD_EXP_FLAGS=$f python scripts/cfg5_costs.py 2>/dev/null | tail -1)"; done 2>&1 | tee gpurun_out/exp_scatter.log

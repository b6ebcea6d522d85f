#!/bin/bash
# round 5, call i: A/B of the classify pass (loads in flight, non-temporal) at the dense configs[4] boundary
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_primal.py -q -m gpu -k "routes_every_scene" 2>&1 | grep -E "passed|failed|^FAILED" | cut -c1-200
EXTRA="--config 4 --mode dense --no-companions" STEPS=100 bash tools/ab_bench.sh 2>&1 | tee $O/r05_i_ab.txt

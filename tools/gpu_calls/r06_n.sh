#!/bin/bash
# round 6, call n: prologue loads issued before the first wait (forward, dense loader, both backwards) - parity, then A/B split / one-kernel backward
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_headline_parity.py tests/test_hip_parity.py tests/test_hip_step_backward.py -m gpu -x -q 2>&1 | tail -4
for pass in 1 2; do STEPS=200 bash tools/ab_bench.sh; done 2>&1 | tee $O/r06_ab_prologue.txt
EXTRA="--mode dense" STEPS=100 bash tools/ab_bench.sh 2>&1 | tee -a $O/r06_ab_prologue.txt
EXTRA="--bwd physical" STEPS=200 bash tools/ab_bench.sh 2>&1 | tee -a $O/r06_ab_prologue.txt

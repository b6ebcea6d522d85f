#!/bin/bash
# round 6, call k: dense backward as solve + streaming kernel - parity of the dense backward cases, then A/B against the one-kernel form
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_headline_parity.py tests/test_hip_parity.py tests/test_hip_quad.py -m gpu -x -q 2>&1 | tail -4
cp lcp_physics_amd/csrc/liblcp_hip.so lcp_physics_amd/csrc/variants/split_nt.so
for pass in 1 2; do STEPS=200 bash tools/ab_bench.sh; done 2>&1 | tee $O/r06_ab_bwd_split.txt
EXTRA="--mode dense" STEPS=100 bash tools/ab_bench.sh 2>&1 | tee -a $O/r06_ab_bwd_split.txt

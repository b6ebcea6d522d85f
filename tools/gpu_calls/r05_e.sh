#!/bin/bash
O=gpurun_out; mkdir -p $O
for v in chain_cmp; do
  echo "== $v"
  LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/$v.so timeout 600 python -m pytest tests/test_hip_primal.py -q -m gpu -s -k "chains_of_joints and 8-2-16" 2>&1 | grep -E "passed|failed|^FAILED|scene" | head -60 | cut -c1-200
done

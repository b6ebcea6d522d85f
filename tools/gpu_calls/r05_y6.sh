#!/bin/bash
cd "$(dirname "$0")/../.."
export LCP_DIAG_DENSEQ=1
for a in "4 4 1024 float64 auto" "2 4 1024 float64 auto" "3 2 1024 float64 auto" "4 2 1024 float32 auto"; do
  timeout 120 python tools/experiments/own_iterate_diag.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
unset LCP_DIAG_DENSEQ
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x 2>&1 | tail -3

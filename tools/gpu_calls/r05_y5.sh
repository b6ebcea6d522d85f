#!/bin/bash
cd "$(dirname "$0")/../.."
export LCP_DIAG_DENSEQ=1
for a in "3 2 1024 float64 auto generic" "4 4 1024 float64 auto generic" "4 2 1024 float32 auto generic" "2 4 1024 float64 auto"; do
  timeout 300 python tools/experiments/own_iterate_diag.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-330
done

#!/bin/bash
cd "$(dirname "$0")/../.."
for a in "3 2 1024 float64 auto generic" "4 4 1024 float64 generic" "5 2 256 float32 generic" "2 2 1024 float64 auto generic"; do
  timeout 300 python tools/experiments/own_iterate_diag.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
timeout 900 python -m pytest tests -q -m gpu -x -k "generic" 2>&1 | tail -3

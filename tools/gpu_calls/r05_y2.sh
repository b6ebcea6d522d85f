#!/bin/bash
cd "$(dirname "$0")/../.."
for a in "4 4 1024 float64 auto generic" "2 4 1024 float64 auto generic" "3 2 1024 float64 auto generic" "4 1 1024 float64 auto generic" "4 2 1024 float32 auto big"; do
  timeout 300 python tools/experiments/own_iterate_diag.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-330
done

#!/bin/bash
cd "$(dirname "$0")/../.."
for a in "5 4 1024 float32 auto big" "6 3 1024 float32 auto big" "9 2 1024 float32 auto big" "5 4 1024 float64 auto"; do
  timeout 200 python tools/experiments/own_iterate_diag.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-330
done

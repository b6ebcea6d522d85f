#!/bin/bash
cd "$(dirname "$0")/../.."
for a in "5 4 256 float32 auto big generic" "5 2 256 float32 auto big generic" "8 2 128 float32 auto big" "4 4 256 float32 auto big generic"; do
  timeout 300 python tools/experiments/own_iterate_diag.py $a 2>&1 | grep -v amdgpu.ids | cut -c1-330
done

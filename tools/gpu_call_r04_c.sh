#!/bin/bash
O=gpurun_out; mkdir -p $O
D=tools/experiments/bwd_outliers.py
{
echo "=== configs2 pinned (fused step + body-space dense backward)"; timeout 300 python $D stack 4096 4 1236 pinned 2>&1 | grep -v amdgpu.ids
echo "=== configs2 dense boundary, contact space"; timeout 300 python $D stack 4096 4 1236 pinned dense big 2>&1 | grep -v amdgpu.ids
echo "=== configs1 coupled rows (ALG = 1)"; timeout 300 python $D stack 1024 2 1236 coupled 2>&1 | grep -v amdgpu.ids
echo "=== configs1 pinned"; timeout 300 python $D stack 1024 2 1236 pinned 2>&1 | grep -v amdgpu.ids
for v in floor11 floor13 refine3; do echo "=== configs2 pinned, variant $v"; LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/$v.so timeout 300 python $D stack 4096 4 1236 pinned 2>&1 | grep -v amdgpu.ids; done
} > $O/r04_bwd_outliers.txt 2>&1
cat $O/r04_bwd_outliers.txt | cut -c1-420

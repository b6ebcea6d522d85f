#!/bin/bash
O=gpurun_out; mkdir -p $O
D=tools/experiments/bwd_outliers.py
{
for v in main n9floor9 n9floor11 n9floor13; do
  L=$PWD/lcp_physics_amd/csrc/variants/$v.so; [ $v = main ] && L=$PWD/lcp_physics_amd/csrc/liblcp_hip.so
  echo "=== configs1 pinned, $v"; LCP_HIP_LIB=$L timeout 300 python $D stack 1024 2 1236 pinned 2>&1 | grep -v amdgpu.ids | grep -v "^scene" 
  echo "=== configs1 coupled, $v"; LCP_HIP_LIB=$L timeout 300 python $D stack 1024 2 1236 coupled 2>&1 | grep -v amdgpu.ids | grep -v "^scene"
done
echo "=== configs2 pinned main (floor 1e-12)"; timeout 300 python $D stack 4096 4 1236 pinned 2>&1 | grep -v amdgpu.ids | head -6
echo "=== configs2 scaled rows main"; timeout 300 python $D stack 4096 4 1236 scaled 2>&1 | grep -v amdgpu.ids | head -6
echo "=== configs2 dense main"; timeout 300 python $D stack 4096 4 1236 pinned dense 2>&1 | grep -v amdgpu.ids | head -6
} > $O/r04_bwd_outliers_2.txt 2>&1
cat $O/r04_bwd_outliers_2.txt | cut -c1-330
timeout 1200 python -m pytest tests -q -m gpu > $O/r04_gputests.log 2>&1; echo "suite rc=$?"; tail -15 $O/r04_gputests.log | cut -c1-400

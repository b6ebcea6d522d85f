#!/bin/bash
# round 6, call d: partial dumps - which of them leave the failure in place
O=gpurun_out; mkdir -p $O
for v in d1 d2 d3 d4 d5 d6; do
  LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/$v.so timeout 300 python tools/experiments/chain_iterates.py dump $v 2>&1 | grep -v amdgpu.ids | tail -3
done
( for v in d1 d2 d3 d4 d5 d6; do python tools/experiments/chain_iterates.py compare $v dhot --dbg; done ) 2>&1 | cut -c1-300 | tee $O/r06_chain_iterates_d.txt

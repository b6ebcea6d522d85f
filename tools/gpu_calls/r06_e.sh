#!/bin/bash
# round 6, call e: assembly-level register dumps of the failing build (and the re-assembled original as the control)
O=gpurun_out; mkdir -p $O
LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/asm_orig.so timeout 300 python tools/experiments/chain_rootcause.py 2>&1 | grep -E "RESULT"
for v in asm_inst1; do
  LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/$v.so timeout 300 python tools/experiments/chain_regdump.py $v 2>&1 | grep -v amdgpu.ids | tail -5
done

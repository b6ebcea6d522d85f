#!/bin/bash
# round 6, call f: the failing build with its two spill instructions moved behind the exec restore (assembly patch, nothing else changed)
for v in asm_orig asm_fixed; do
  echo "== $v"
  LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/$v.so timeout 300 python tools/experiments/chain_rootcause.py 2>&1 | grep -E "^case|RESULT" | cut -c1-200
done 2>&1 | tee gpurun_out/r06_chain_asm_fix.txt

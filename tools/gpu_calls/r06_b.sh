#!/bin/bash
# round 6, call b: where the iterates of the failing build first part from the passing ones
O=gpurun_out; mkdir -p $O
for v in fast hot exact snop; do
  LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/$v.so timeout 300 python tools/experiments/chain_iterates.py dump $v 2>&1 | tail -3
done
( python tools/experiments/chain_iterates.py compare hot exact; python tools/experiments/chain_iterates.py compare fast hot; python tools/experiments/chain_iterates.py compare snop fast ) 2>&1 | tee $O/r06_chain_iterates_b.txt

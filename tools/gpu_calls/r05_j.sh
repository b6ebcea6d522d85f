#!/bin/bash
# round 5, call j: upper bound of block pivots in the config-5 LU (tools/experiments/block_pivot_bound.py)
O=gpurun_out; mkdir -p $O
for rep in 1 2; do for v in pivots_every1 pivots_every3 pivots_every30; do
  LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/$v.so timeout 300 python tools/experiments/block_pivot_bound.py 2>/dev/null | tee -a $O/r05_j_block_pivot_bound.txt
done; done

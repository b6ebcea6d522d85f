#!/bin/bash
# round 6, call c: the dump of the initialisation pass (factors, first solve) from the failing and the passing builds
O=gpurun_out; mkdir -p $O
for v in dfast dhot dexact; do
  LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/$v.so timeout 300 python tools/experiments/chain_iterates.py dump $v 2>&1 | grep -v amdgpu.ids | tail -3
done
( python tools/experiments/chain_iterates.py compare dhot dexact --dbg; python tools/experiments/chain_iterates.py compare dfast dhot --dbg ) 2>&1 | cut -c1-400 | tee $O/r06_chain_iterates_c.txt

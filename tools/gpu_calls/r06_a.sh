#!/bin/bash
# round 6, call a: the <56>-column chain instantiation with reciprocal step lengths - which build flag / source form moves the failure
O=gpurun_out; mkdir -p $O
for v in fast exact snop wait0 O1 O2 nopostsched nomisched cold hot noexpect; do
  echo "== $v"
  LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/$v.so timeout 300 python tools/experiments/chain_rootcause.py 2>&1 | grep -E "^case|RESULT|Error|error" | cut -c1-260
done 2>&1 | tee $O/r06_chain_variants_a.txt
timeout 300 python bench.py > $O/r06_bench_start.json 2> $O/r06_bench_start.err; tail -1 $O/r06_bench_start.json | cut -c1-300

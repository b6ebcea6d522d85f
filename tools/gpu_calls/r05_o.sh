#!/bin/bash
O=gpurun_out; mkdir -p $O
STEPS=200 EXTRA="--no-companions" bash tools/ab_bench.sh 2>&1 | tee $O/r05_o_ab.txt
STEPS=200 EXTRA="--no-companions" bash tools/ab_bench.sh 2>&1 | tee -a $O/r05_o_ab.txt
LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/refine_skip.so timeout 600 python tools/experiments/headline_dump.py dump $O/r05_dump_skip configs2_4096x16 configs2_4096x4_one_point > $O/r05_o_dump.log 2>&1; echo "dump rc=$?"

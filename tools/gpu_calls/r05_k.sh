#!/bin/bash
# round 5, call k: dense backward with one refinement step instead of two (LCP_Q_BWD_REFINE): speed + kernel outputs for the offline oracle report
O=gpurun_out; mkdir -p $O
STEPS=200 bash tools/ab_bench.sh 2>&1 | tee $O/r05_k_ab.txt
STEPS=200 bash tools/ab_bench.sh 2>&1 | tee -a $O/r05_k_ab.txt
LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/refine1.so timeout 600 python tools/experiments/headline_dump.py dump $O/r05_dump_refine1 configs2_4096x16 configs2_4096x4_one_point > $O/r05_k_dump.log 2>&1; echo "dump rc=$?"

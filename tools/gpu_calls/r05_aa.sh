#!/bin/bash
# round 5, call aa: one Newton step instead of two behind v_rcp_f64 (headline unit only): A/B + the kernel outputs for the offline parity report
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
STEPS=20 bash tools/ab_bench.sh > gpurun_out/r05_aa_ab.txt 2>&1
STEPS=20 bash tools/ab_bench.sh >> gpurun_out/r05_aa_ab.txt 2>&1
cat gpurun_out/r05_aa_ab.txt
LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/nr1.so timeout 600 python tools/experiments/headline_dump.py dump gpurun_out/r05_dump_nr1 configs2_4096x16 configs2_4096x4_one_point > gpurun_out/r05_aa_dump.log 2>&1; echo "dump rc=$?"

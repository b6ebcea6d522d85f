#!/bin/bash
# round 5, call b: A/B of the reciprocal step lengths (LCP_Q_RCP_STEP) on the headline + kernel outputs of the new default for the offline oracle report
O=gpurun_out; mkdir -p $O
STEPS=200 bash tools/ab_bench.sh 2>&1 | tee $O/r05_b_ab.txt
STEPS=200 bash tools/ab_bench.sh 2>&1 | tee -a $O/r05_b_ab.txt
timeout 600 python tools/experiments/headline_dump.py dump $O/r05_dump_rcp configs1_1024x8 configs2_4096x16 configs2_4096x16_count configs1_1024x8_count configs2_4096x16_dense configs2_4096x8_two_points > $O/r05_b_dump.log 2>&1; echo "dump rc=$?"

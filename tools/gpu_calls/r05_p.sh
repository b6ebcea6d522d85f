#!/bin/bash
# round 5, call p: dense-boundary classification with 16-byte loads (lcp_classify_wave) - A/B + the tests that cover it
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
EXTRA="--mode dense" STEPS=20 bash tools/ab_bench.sh > gpurun_out/r05_p_ab.txt 2>&1
EXTRA="--mode dense" STEPS=20 bash tools/ab_bench.sh >> gpurun_out/r05_p_ab.txt 2>&1
cat gpurun_out/r05_p_ab.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "dense or classif or general or lcp_function or routing or parity" 2>&1 | tail -5

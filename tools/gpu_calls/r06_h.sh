#!/bin/bash
# round 6, call h: index-set accounting - every headline / dense parity case with the oracle-stability fields; bench parity objects for configs 1..4
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests/test_hip_headline_parity.py -m gpu -q -s -k "timed_kernel_against or dense_boundary" 2>&1 | grep -E "^headline parity|passed|failed|Error|assert" > $O/r06_headline_parity_h.txt
tail -3 $O/r06_headline_parity_h.txt | cut -c1-300
for c in 1 2 3 4; do
  timeout 600 python bench.py --config $c > $O/r06_bench_config${c}_h.json 2> $O/r06_bench_config${c}_h.err; tail -1 $O/r06_bench_config${c}_h.json | cut -c1-150
done

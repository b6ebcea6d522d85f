#!/bin/bash
# round 5, call l: whole GPU suite at HEAD (one refinement step in the body-space backward, device SIMD count, dense_step guards) + dumps for the offline report
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r05_l_gputests.log 2>&1; echo "suite rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" $O/r05_l_gputests.log | tail -8 | cut -c1-400
timeout 600 python tools/experiments/headline_dump.py dump $O/r05_dump_head configs1_1024x8 configs2_4096x16 configs2_4096x16_dense configs4_4096x64_pile configs4_4096x64_pile_dense > $O/r05_l_dump.log 2>&1; echo "dump rc=$?"
run() { name=$1; shift; timeout 300 "$@" > $O/r05_$name.json 2> $O/r05_$name.err; tail -1 $O/r05_$name.json | cut -c1-200; }
run l_bench_fused python bench.py --no-cpu-baseline
run l_bench_dense python bench.py --mode dense --no-cpu-baseline
run l_bench_physical python bench.py --bwd physical --no-cpu-baseline

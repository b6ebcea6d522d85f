#!/bin/bash
# round 5, call c: the whole GPU suite on the reciprocal-step kernels (quad, solo, primal), bench lines of configs[1] / [2] / [4], dumps for the offline report
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/r05_c_gputests.log 2>&1; echo "suite rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" $O/r05_c_gputests.log | tail -5 | cut -c1-400
run() { name=$1; shift; timeout 300 "$@" > $O/r05_$name.json 2> $O/r05_$name.err; tail -1 $O/r05_$name.json | cut -c1-200; }
run c_bench_fused python bench.py --no-cpu-baseline
run c_bench_config2 python bench.py --config 1 --no-cpu-baseline
run c_bench_config5 python bench.py --config 4 --no-cpu-baseline
run c_bench_config5_dense python bench.py --config 4 --mode dense --no-cpu-baseline
run c_bench_dense python bench.py --mode dense --no-cpu-baseline
timeout 600 python tools/experiments/headline_dump.py dump $O/r05_dump_rcp configs1_1024x8 configs1_1024x8_count configs2_4096x4_one_point configs4_4096x64_pile configs4_4096x64_pile_dense > $O/r05_c_dump.log 2>&1; echo "dump rc=$?"

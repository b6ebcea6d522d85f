#!/bin/bash
# round 5, call d: exact cold path of the reciprocal step lengths (all kernels), A/B of the flat bookkeeping, benches
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r05_d_gputests.log 2>&1; echo "suite rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" $O/r05_d_gputests.log | tail -8 | cut -c1-400
STEPS=200 bash tools/ab_bench.sh 2>&1 | tee $O/r05_d_ab.txt
STEPS=200 bash tools/ab_bench.sh 2>&1 | tee -a $O/r05_d_ab.txt
run() { name=$1; shift; timeout 300 "$@" > $O/r05_$name.json 2> $O/r05_$name.err; tail -1 $O/r05_$name.json | cut -c1-200; }
run d_bench_config2 python bench.py --config 1 --no-cpu-baseline
run d_bench_config5 python bench.py --config 4 --no-cpu-baseline
run d_bench_config5_dense python bench.py --config 4 --mode dense --no-cpu-baseline
timeout 600 python tools/experiments/headline_dump.py dump $O/r05_dump_rcp configs1_1024x8 configs2_4096x16 configs4_4096x64_pile > $O/r05_d_dump.log 2>&1; echo "dump rc=$?"

#!/bin/bash
# round 6, call g: the library built through compile_unit.sh (spill code moved behind exec restores; chain unit on reciprocal step lengths):
# the whole GPU suite, then the headline bench and config 5
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/r06_gputests_g.txt
timeout 300 python bench.py > $O/r06_bench_fused_g.json 2> $O/r06_bench_fused_g.err; tail -1 $O/r06_bench_fused_g.json | cut -c1-200
timeout 300 python bench.py --config 4 --no-cpu-baseline > $O/r06_bench_config5_g.json 2> $O/r06_bench_config5_g.err; tail -1 $O/r06_bench_config5_g.json | cut -c1-200

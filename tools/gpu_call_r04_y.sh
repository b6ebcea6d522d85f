#!/bin/bash
O=gpurun_out
timeout 1800 python -m pytest tests -q -m gpu > $O/r04_gputests.log 2>&1; echo "suite rc=$?"; grep -E "^FAILED|passed|failed" $O/r04_gputests.log | tail -5 | cut -c1-250
run() { name=$1; shift; timeout 300 "$@" > $O/r04_$name.json 2> $O/r04_$name.err; tail -1 $O/r04_$name.json | cut -c1-140; }
run bench_fused python bench.py
run bench_driver_form python bench.py --steps 20 --warmup 5
run bench_dense python bench.py --mode dense --cpu-budget 3
run bench_fused_physical_bwd python bench.py --bwd physical --no-cpu-baseline
run bench_config2_fwd_only python bench.py --config 1 --no-cpu-baseline
run bench_config4_on_1gpu python bench.py --batch 32768 --no-cpu-baseline
run bench_fused_8contacts python bench.py --pts 2 --no-cpu-baseline
run bench_world python tools/bench_world.py --cpu-scenes 2
run bench_world_graph python tools/bench_world.py --cpu-scenes 0 --graph
run bench_world_6bodies python tools/bench_world.py --nbox 5 --box 40 --cpu-scenes 0
run batch_curve_4box python tools/bench_batch_curve.py 4
run batch_curve_2box python tools/bench_batch_curve.py 2
EXTRA="" bash tools/profile_all.sh r04 > $O/r04_profile_all.log 2>&1
EXTRA="--mode dense" bash tools/profile_all.sh r04dense > $O/r04_profile_dense.log 2>&1
LCP_HIP_LIB=$PWD/tools/liblcp_quadprof.so timeout 200 python tools/gpu_phase_profile_quad.py 4096 4 > $O/r04_quad_phase_profile.txt 2>&1
LCP_HIP_LIB=$PWD/tools/liblcp_quadprof.so timeout 200 python tools/gpu_phase_profile_quad.py 32768 4 >> $O/r04_quad_phase_profile.txt 2>&1
head -4 $O/prof_r04_trace.summary.txt | cut -c1-130

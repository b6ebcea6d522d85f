#!/bin/bash
O=gpurun_out
timeout 1800 python -m pytest tests -q -m gpu > $O/r04_gputests.log 2>&1; echo "suite rc=$?"; grep -E "^FAILED|passed|failed" $O/r04_gputests.log | tail -5 | cut -c1-250
run() { name=$1; shift; timeout 300 "$@" > $O/r04_$name.json 2> $O/r04_$name.err; tail -1 $O/r04_$name.json | cut -c1-160; }
run bench_config2_fwd_only python bench.py --config 1 --no-cpu-baseline
run bench_config5 python bench.py --config 4
run bench_midsize_24 python tools/bench_midsize.py 6 4
run bench_midsize_32 python tools/bench_midsize.py 8 4
run bench_world_11bodies python tools/bench_world.py --nbox 10 --box 24 --maxc 32 --cpu-scenes 0
run batch_curve_2box python tools/bench_batch_curve.py 2
bash tools/profile_config5.sh r04 > $O/r04_profile_config5.log 2>&1; tail -3 $O/r04_profile_config5.log | cut -c1-200

#!/bin/bash
# round 6: full GPU suite after the generic step / post-stab backward; a 20-body bench line (fwd + physical bwd on the generic kernels);
# configs[4] dense line (its backward after the prologue hoists)
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_u_gputests.txt
timeout 300 python bench.py --nbox 19 --pts 2 --batch 1024 --no-cpu-baseline 2>gpurun_out/r06_u_b20.err | tail -1 > gpurun_out/r06_u_bench_20bodies.json
timeout 300 python bench.py --config 4 --mode dense --no-cpu-baseline 2>gpurun_out/r06_u_c4.err | tail -1 > gpurun_out/r06_u_bench_c4_dense.json
tail -5 gpurun_out/r06_u_gputests.txt; cut -c1-600 gpurun_out/r06_u_bench_20bodies.json; tail -3 gpurun_out/r06_u_b20.err

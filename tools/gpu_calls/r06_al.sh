#!/bin/bash
# round 6: the two remaining lines on the stamped counters - eight ranks on one device, the 20-body step with its physical backward
cd /root/repo; mkdir -p gpurun_out
timeout 900 python bench.py --gpus 8 --share-devices --no-cpu-baseline > gpurun_out/r06_bench_8ranks_one_device.json 2> gpurun_out/r06_bench_8ranks_one_device.err; tail -1 gpurun_out/r06_bench_8ranks_one_device.json | cut -c1-300
timeout 300 python bench.py --nbox 19 --pts 2 --batch 1024 --bwd physical --no-cpu-baseline > gpurun_out/r06_bench_step_20bodies_physical.json 2>/dev/null; tail -1 gpurun_out/r06_bench_step_20bodies_physical.json | cut -c1-300
timeout 600 python tools/bench_world.py --batch 1024 --nbox 19 --maxc 48 --steps 20 --settle 10 --record 4 --cpu-scenes 0 --post-stab > gpurun_out/r06_bench_world_20bodies_post_stab.json 2>/dev/null; tail -1 gpurun_out/r06_bench_world_20bodies_post_stab.json | cut -c1-200

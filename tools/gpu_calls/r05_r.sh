#!/bin/bash
# round 5, call r: the world with post-stabilisation, strict (the retry loop of world.py:88-101 has no floor: scenes whose corrected pose
# still penetrates halve dt to the cap of 64 trials) and non-strict (stops at dt / 4)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python tools/bench_world.py --cpu-scenes 0 --post-stab > gpurun_out/r05_bench_world_post_stab.json 2>/dev/null
python tools/bench_world.py --cpu-scenes 0 --post-stab --no-strict > gpurun_out/r05_bench_world_post_stab_nonstrict.json 2>/dev/null
python tools/bench_world.py --cpu-scenes 0 --no-strict > gpurun_out/r05_bench_world_nonstrict.json 2>/dev/null
python tools/bench_world.py --cpu-scenes 0 --post-stab --no-strict --graph > gpurun_out/r05_bench_world_post_stab_nonstrict_graph.json 2>/dev/null
cat gpurun_out/r05_bench_world_post_stab*.json gpurun_out/r05_bench_world_nonstrict.json | cut -c1-700

#!/bin/bash
# round 6: soak of the big-scene path - 20 bodies, 300 forward steps with detection, then 16-step recorded roll-outs (finite gradients asserted by the tool)
cd /root/repo; mkdir -p gpurun_out
timeout 900 python tools/bench_world.py --batch 512 --nbox 19 --maxc 48 --steps 300 --settle 10 --record 16 --record-reps 2 --cpu-scenes 0 2>gpurun_out/r06_soak.err | tail -1 > gpurun_out/r06_soak_world_20bodies.json
cut -c1-1400 gpurun_out/r06_soak_world_20bodies.json; tail -2 gpurun_out/r06_soak.err

#!/bin/bash
# round 6: when do the slow stretches of a run happen (tools/experiments/step_hiccups.py), three processes each
cd /root/repo; mkdir -p gpurun_out
{ for rep in 1 2 3; do timeout 200 python tools/experiments/step_hiccups.py --config 3 --seconds 3 2>/dev/null | tail -1; done
  for rep in 1 2; do timeout 200 python tools/experiments/step_hiccups.py --config 4 --seconds 3 2>/dev/null | tail -1; done
  for rep in 1 2; do timeout 200 python tools/experiments/step_hiccups.py --seconds 2 --group 50 2>/dev/null | tail -1; done
} | tee gpurun_out/r06_step_hiccups.txt

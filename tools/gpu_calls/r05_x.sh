#!/bin/bash
# the GPU suite at HEAD (+ the slowest tests)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --durations=30 2>&1 | tail -50 | cut -c1-300 > gpurun_out/r05_gputests.txt
cat gpurun_out/r05_gputests.txt

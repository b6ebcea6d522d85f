#!/bin/bash
# round 6: the GPU suite at HEAD with the slowest tests, smoke()
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --durations=25 2>&1 | tail -45 > gpurun_out/r06_gputests.txt
tail -4 gpurun_out/r06_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

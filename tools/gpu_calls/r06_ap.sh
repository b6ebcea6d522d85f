#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --durations=25 2>&1 | tail -45 > gpurun_out/r06_gputests.txt
tail -6 gpurun_out/r06_gputests.txt

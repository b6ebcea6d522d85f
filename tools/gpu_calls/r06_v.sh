#!/bin/bash
# round 6: the tail of the GPU suite after the one expectation that changed (the generic path now has a step backward)
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_step_backward.py $(ls tests/test_[i-z]*.py) -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06_v_tail.txt
cat gpurun_out/r06_v_tail.txt

#!/bin/bash
# round 6: the generic step backward - its tests first, then the suites that touch the step path
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_dense_step.py tests/test_hip_reentrancy.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r06_t_dense.txt
cat gpurun_out/r06_t_dense.txt

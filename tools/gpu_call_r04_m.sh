#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_dense_step.py -q -s -x > $O/r04_dense_step_tests.log 2>&1; echo "rc=$?"; tail -40 $O/r04_dense_step_tests.log | cut -c1-400

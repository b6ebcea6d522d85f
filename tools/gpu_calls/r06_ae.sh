#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_dense_step.py -m gpu -q -s -k "beyond_64" 2>&1 | tail -25

#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_headline_parity.py tests/test_hip_parity.py tests/test_hip_step_backward.py -q -m gpu -x -k "not against_oracle_at_metric_sizes and not nonredundant" 2>&1 | tail -40 | cut -c1-600

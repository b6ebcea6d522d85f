#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_primal.py -q -m gpu -k "four_scenes_per_wave" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-400 | head -30

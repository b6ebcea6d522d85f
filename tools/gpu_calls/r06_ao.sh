#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_hip_dense_step.py -m gpu -q -s -k "twenty_bodies" 2>&1 | grep "per scene:\|step, worst\|fused step,\|passed\|failed\|Error" | cut -c1-700

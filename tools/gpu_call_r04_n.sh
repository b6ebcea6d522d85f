#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_contacts.py -q -s -x -k "joint or rollout or differentiable" > $O/r04_joint_anchor_tests.log 2>&1; echo "rc=$?"; grep -v "^$" $O/r04_joint_anchor_tests.log | tail -25 | cut -c1-300

#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests/test_hip_headline_parity.py -m gpu -q -s -k "configs1" 2>&1 | grep -E "^headline parity|passed|failed|Error|assert" > $O/r06_headline_parity_i.txt
tail -3 $O/r06_headline_parity_i.txt | cut -c1-300

#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -s -k "converged_solves" 2>&1 | grep -v "^$" | tail -20 | cut -c1-330

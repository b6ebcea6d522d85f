#!/bin/bash
# round 5, call s: the fp64-I/O operator test with the every-scene gates (no same-iterate filter)
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -s -k "fp64_io_takes" 2>&1 | grep -v "^$" | tail -15

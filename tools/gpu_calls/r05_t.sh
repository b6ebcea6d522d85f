#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 600 python tools/experiments/fp64_io_backward_diag.py auto 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-400
bash tools/gpu_calls/r05_s.sh

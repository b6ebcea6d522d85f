#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "engine or plugin or world or Engine" 2>&1 | tail -3
python tools/experiments/engine_latency.py > gpurun_out/r05_engine_latency.json 2>/dev/null; cat gpurun_out/r05_engine_latency.json | cut -c1-1200

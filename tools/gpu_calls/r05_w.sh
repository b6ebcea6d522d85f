#!/bin/bash
cd "$(dirname "$0")/../.."
bash tools/gpu_calls/r05_t.sh 2>&1 | grep -v "^$" | tail -14 | cut -c1-400
bash tools/gpu_calls/r05_v.sh

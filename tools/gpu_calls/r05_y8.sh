#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_headline_parity.py -q -m gpu -x -s -k "two_point_shape" 2>&1 | grep "^headline parity" > gpurun_out/r05_two_points_report.txt
python - <<'PY'
import json
s=open("gpurun_out/r05_two_points_report.txt").read()
d=json.loads(s[s.index("{"):])
for k,v in d.items():
    if k.startswith("bwd") or k.startswith("iters") or k.startswith("index"): print(k, v)
PY

#!/bin/bash
# round 5, call u: the contact-space backward's tiny-pivot fallback + the own-iterate gates: headline parity cases, operator parity, dense / contact-space benches
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_headline_parity.py tests/test_hip_parity.py tests/test_hip_step_backward.py -q -m gpu -x -s 2>&1 | grep -o "headline parity [a-z0-9_]*\|\"bwd_own_iterate_[a-z_]*\": [^,]*\|passed.*\|failed.*\|Error.*\|oracle backward at the kernel.*" | cut -c1-300 > gpurun_out/r05_u_tests.txt
tail -60 gpurun_out/r05_u_tests.txt
for a in "--mode dense" "--mode dense --contact-space"; do python bench.py $a --cpu-budget 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', d['value'], d['roofline']['fwd_ms'], d['roofline']['bwd_ms'], {k:v for k,v in (d.get("parity") or {}).items() if "own_iterate" in k})"; done

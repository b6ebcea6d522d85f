#!/bin/bash
O=gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_lcpfunction.py tests/test_hip_reentrancy.py -q -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_hip_headline_parity.py -q -x -k dense 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --mode dense --no-cpu-baseline --no-companions 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('dense: %.2f M  ms/step %.4f  fwd %.4f ms  bwd %.4f ms' % (j['value']/1e6, j['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"; done

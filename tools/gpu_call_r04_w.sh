#!/bin/bash
bench() { timeout 300 python bench.py --config 4 --no-cpu-baseline --no-companions 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$1: %.2f M  fwd %.4f ms  bwd %.4f ms' % (j['value']/1e6, r['fwd_ms'], r['bwd_ms']))"; }
for i in 1 2; do bench "HEAD           "; LCP_HIP_LIB=$PWD/tools/liblcp_exp_punr.so bench "passes unrolled"; done
python tools/bench_midsize.py 6 4 2>/dev/null | tail -1; LCP_HIP_LIB=$PWD/tools/liblcp_exp_punr.so python tools/bench_midsize.py 6 4 2>/dev/null | tail -1
LCP_HIP_LIB=$PWD/tools/liblcp_exp_punr.so timeout 600 python -m pytest tests/test_hip_primal.py -q -x 2>&1 | tail -2
LCP_HIP_LIB=$PWD/tools/liblcp_exp_punr.so timeout 600 python -m pytest tests/test_hip_headline_parity.py -q -x -k configs4 2>&1 | tail -2

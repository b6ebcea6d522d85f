#!/bin/bash
bench() { timeout 300 python bench.py --no-cpu-baseline --no-companions $2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$1 %s: %.2f M  ms/step %.4f  fwd %.4f ms  bwd %.4f ms' % ('$2', j['value']/1e6, j['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"; }
for i in 1 2 3; do bench "HEAD        "; LCP_HIP_LIB=$PWD/tools/liblcp_exp_nt.so bench "nt stores dF"; done

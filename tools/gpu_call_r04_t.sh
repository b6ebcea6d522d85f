#!/bin/bash
bench() { timeout 300 python bench.py --config 4 --no-cpu-baseline --no-companions 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; p=j.get('parity',{})
print('$1: %.2f M  fwd %.4f ms  bwd %.4f ms  err_x %s iters_differ %s unmasked %s' % (j['value']/1e6, r['fwd_ms'], r['bwd_ms'], p.get('fwd_err_x_max'), p.get('iters_differ_frac'), p.get('index_set_mismatches_unmasked')))"; }
for i in 1 2; do bench "HEAD          "; LCP_HIP_LIB=$PWD/tools/liblcp_exp_rcp1.so bench "rcp 1 Newton  "; done

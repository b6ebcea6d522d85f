#!/bin/bash
# VERDICT r03 item 4, measured: the headline forward with 4 (HEAD), 2 and 1 live scenes per wavefront at 4096 scenes
bench() { timeout 300 python bench.py --no-cpu-baseline --no-companions $2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$1 %s: %.2f M  ms/step %.4f  fwd %.4f ms  bwd %.4f ms  parity err_x %s iters_equal %s' % ('$2', j['value']/1e6, j['ms_per_step'], r['fwd_ms'], r['bwd_ms'], j.get('parity',{}).get('fwd_err_x_max'), j.get('parity',{}).get('iters_equal')))"; }
for i in 1 2; do
bench "4 scenes per wave, one wave per SIMD (HEAD)            "
LCP_HIP_LIB=$PWD/tools/liblcp_exp_spw2.so bench "2 scenes per wave, two waves per SIMD (lean kernel)     "
LCP_HIP_LIB=$PWD/tools/liblcp_exp_spw1.so bench "1 scene per wave, four waves of work per SIMD (two live)"
done
bench "4 scenes per wave, 32768 scenes (lean kernel, 2 live)  " "--batch 32768"
LCP_HIP_LIB=$PWD/tools/liblcp_exp_spw2.so bench "2 scenes per wave, 32768 scenes                        " "--batch 32768"

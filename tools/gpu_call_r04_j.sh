#!/bin/bash
O=gpurun_out; mkdir -p $O
V=$PWD/lcp_physics_amd/csrc/variants
{
for v in main separate_fallbacks; do
  L=$V/$v.so; [ $v = main ] && L=$PWD/lcp_physics_amd/csrc/liblcp_hip.so
  echo "== $v"
  LCP_HIP_LIB=$L timeout 300 python tools/experiments/general_dense_time.py 2>&1 | grep -v amdgpu
  for rep in 1 2; do LCP_HIP_LIB=$L timeout 300 python bench.py --mode dense --no-cpu-baseline --no-companions 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('dense 4096 x 16 (all scenes contact-structured): %.3f M fwd+bwd  ms/step %.4f  fwd %.4f ms  bwd %.4f ms' % (j['value']/1e6, j['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"; done
done
} > $O/r04_ab_fallbacks.txt 2>&1
cat $O/r04_ab_fallbacks.txt
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_headline_parity.py tests/test_hip_step_backward.py -q -m gpu -x > $O/r04_fallback_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r04_fallback_tests.log

#!/bin/bash
O=gpurun_out; mkdir -p $O
V=$PWD/lcp_physics_amd/csrc/variants
timeout 900 python -m pytest tests/test_hip_primal.py tests/test_hip_step_backward.py tests/test_hip_headline_parity.py -q -m gpu -x -k "not dense" > $O/r04_grid_tests_main.log 2>&1; echo "main (grid v2 in the backward kernels) rc=$?"; tail -3 $O/r04_grid_tests_main.log
LCP_HIP_LIB=$V/gridfwd.so timeout 900 python -m pytest tests/test_hip_primal.py tests/test_hip_step_backward.py tests/test_hip_headline_parity.py -q -m gpu -x -k "not dense" > $O/r04_grid_tests_fwd.log 2>&1; echo "gridfwd (forward too) rc=$?"; tail -3 $O/r04_grid_tests_fwd.log
{
for v in main gridv1 gridfwd; do
  L=$V/$v.so; [ $v = main ] && L=$PWD/lcp_physics_amd/csrc/liblcp_hip.so
  for rep in 1 2; do
  LCP_HIP_LIB=$L timeout 300 python bench.py --config 4 --no-cpu-baseline --no-companions 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-10s config 4: %.3f M fwd+bwd  ms/step %.4f  fwd %.4f ms (%.2f M fwd only)  bwd %.4f ms' % ('$v', j['value']/1e6, j['ms_per_step'], r['fwd_ms'], 4096/r['fwd_ms']/1e3, r['bwd_ms']))"
  done
done
for v in rowfwd_prof gridfwd_prof; do echo "== phases $v"; LCP_HIP_LIB=$V/$v.so timeout 200 python tools/config5_phases.py 4096 2>&1 | grep -v amdgpu; done
# other sizes on the same kernels (24 / 32 columns)
for v in main gridfwd; do
  L=$V/$v.so; [ $v = main ] && L=$PWD/lcp_physics_amd/csrc/liblcp_hip.so
  echo "== midsize $v"; LCP_HIP_LIB=$L timeout 200 python tools/bench_midsize.py 6 4 2>&1 | tail -2; LCP_HIP_LIB=$L timeout 200 python tools/bench_midsize.py 8 4 2>&1 | tail -2
done
} > $O/r04_ab_config5.txt 2>&1
cat $O/r04_ab_config5.txt

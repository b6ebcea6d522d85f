#!/bin/bash
O=gpurun_out; mkdir -p $O
V=$PWD/lcp_physics_amd/csrc/variants
timeout 1500 python -m pytest tests -q -m gpu > $O/r04_gputests.log 2>&1; echo "suite rc=$?"; tail -6 $O/r04_gputests.log | cut -c1-300
{
for v in main cmpsweep; do
  L=$V/$v.so; [ $v = main ] && L=$PWD/lcp_physics_amd/csrc/liblcp_hip.so
  for rep in 1 2; do
  LCP_HIP_LIB=$L timeout 300 python bench.py --config 4 --no-cpu-baseline --no-companions 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-10s config 4: %.3f M fwd+bwd  ms/step %.4f  fwd %.4f ms (%.2f M fwd only)  bwd %.4f ms' % ('$v', j['value']/1e6, j['ms_per_step'], r['fwd_ms'], 4096/r['fwd_ms']/1e3, r['bwd_ms']))"
  done
done
echo "== phases execsweep_prof"; LCP_HIP_LIB=$V/execsweep_prof.so timeout 200 python tools/config5_phases.py 4096 2>&1 | grep -v amdgpu
for v in main cmpsweep; do
  L=$V/$v.so; [ $v = main ] && L=$PWD/lcp_physics_amd/csrc/liblcp_hip.so
  echo "== midsize $v"; LCP_HIP_LIB=$L timeout 200 python tools/bench_midsize.py 6 4 2>&1 | tail -1; LCP_HIP_LIB=$L timeout 200 python tools/bench_midsize.py 8 4 2>&1 | tail -1
done
} > $O/r04_ab_config5_d.txt 2>&1
cat $O/r04_ab_config5_d.txt

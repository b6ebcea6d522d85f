#!/bin/bash
O=gpurun_out; mkdir -p $O
for v in bs0 bs2; do
  echo "== $v"; LCP_HIP_LIB=$PWD/tools/liblcp_primalprof_$v.so timeout 200 python tools/config5_phases.py 4096 2>&1 | grep -v amdgpu | cut -c1-260
done
timeout 600 python -m pytest tests/test_hip_primal.py -q -x 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --config 4 --no-cpu-baseline --no-companions 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('config4 %.2f M ms %.4f fwd %.4f bwd %.4f' % (j['value']/1e6, j['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"; done

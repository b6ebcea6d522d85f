#!/bin/bash
O=gpurun_out; mkdir -p $O
bench() { timeout 300 python bench.py --config 4 --no-cpu-baseline --no-companions 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$1 config4 %.2f M ms %.4f fwd %.4f bwd %.4f' % (j['value']/1e6, j['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"; }
for i in 1 2; do bench main; LCP_HIP_LIB=$PWD/tools/liblcp_xrow.so bench xrow; done
for v in primalprof primalprof_xrow; do echo "== $v"; LCP_HIP_LIB=$PWD/tools/liblcp_$v.so timeout 200 python tools/config5_phases.py 4096 2>&1 | grep -v amdgpu | cut -c1-260; done
LCP_HIP_LIB=$PWD/tools/liblcp_xrow.so timeout 600 python -m pytest tests/test_hip_primal.py -q -x 2>&1 | tail -2

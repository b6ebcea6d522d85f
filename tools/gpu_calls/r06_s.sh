#!/bin/bash
# round 6, call s: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) - prologue cycles and the bench line
O=gpurun_out; mkdir -p $O
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v LCP_HIP_LIB=$PWD/tools/liblcp_quadprof.so python tools/gpu_phase_profile_quad.py 4096 4 2>&1 | grep -v amdgpu.ids | head -2
  HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('value %.0f' % d['value'], 'ms_per_step', d['ms_per_step'], 'fwd_ms', r.get('fwd_ms'), 'bwd_ms', r.get('bwd_ms'), 'host_us', d.get('host_us_per_step'))"
done 2>&1 | tee $O/r06_ab_dev_kernarg.txt

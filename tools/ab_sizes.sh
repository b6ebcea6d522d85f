#!/bin/bash
# A/B over batch sizes: bench.py against every library variant under lcp_physics_amd/csrc/variants/ at several --batch values
for lib in lcp_physics_amd/csrc/variants/*.so; do
  n=$(basename $lib .so)
  for B in ${SIZES:-4096 8192 32768}; do
    LCP_HIP_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --batch $B ${EXTRA} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-16s B=%6d  %.2f M/s  fwd %.4f  bwd %.4f  sustained %.2f M/s' % ('$n', $B, j['value']/1e6, r['fwd_ms'], r['bwd_ms'], j['sustained']['value']/1e6))"
  done
done

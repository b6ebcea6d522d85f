#!/bin/bash
# round 6: max-memory-clause on the other timed units (lcp_solo: configs[1]; lcp_primal_pin: configs[4] fused; lcp_primal: configs[4] dense), interleaved A/B;
# then the headline with the adopted Makefile flag (product library)
cd /root/repo; mkdir -p gpurun_out
V=lcp_physics_amd/csrc/variants
run() { # name lib args...
  local n=$1 lib=$2; shift 2
  LCP_HIP_LIB=$lib timeout 300 python bench.py "$@" --no-cpu-baseline --no-companions --sustain 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$n: value %.4g ms/step %.5f fwd %.5f bwd %.5f' % (d['value'], d['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"
}
{
for rep in 1 2; do
  for v in solo_base solo_memcl; do run "$v config1 rep$rep" $V/$v.so --config 1; done
  for v in pin_base pin_memcl; do run "$v config4 rep$rep" $V/$v.so --config 4; done
  for v in primal_base primal_memcl; do run "$v config4-dense rep$rep" $V/$v.so --config 4 --mode dense; done
done
run "product headline" lcp_physics_amd/csrc/liblcp_hip.so
run "product headline" lcp_physics_amd/csrc/liblcp_hip.so
} | tee gpurun_out/r06_y_ab.txt

#!/bin/bash
# round 6: ILP scheduling strategies on the latency-bound units (lcp_solo: 175 registers, one wave per SIMD at 1024 scenes; lcp_primal_pin: configs[4])
cd /root/repo; mkdir -p gpurun_out
V=lcp_physics_amd/csrc/variants
run() { local n=$1 lib=$2; shift 2
  LCP_HIP_LIB=$lib timeout 300 python bench.py "$@" --no-cpu-baseline --no-companions --sustain 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$n: value %.4g ms/step %.5f fwd %.5f bwd %.5f' % (d['value'], d['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"
}
{ for rep in 1 2; do
  for v in solo_base solo_maxilp solo_iterilp; do run "$v config1 rep$rep" $V/$v.so --config 1; done
  for v in pin_base pin_maxilp; do run "$v config4 rep$rep" $V/$v.so --config 4; done
done; } | tee gpurun_out/r06_af_ab.txt

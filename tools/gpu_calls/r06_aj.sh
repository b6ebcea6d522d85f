#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
V=lcp_physics_amd/csrc/variants
run() { local n=$1 lib=$2; shift 2
  LCP_HIP_LIB=$lib timeout 300 python bench.py "$@" --no-cpu-baseline --no-companions --sustain 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$n: value %.4g ms/step %.5f fwd %.5f bwd %.5f' % (d['value'], d['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"
}
{ for rep in 1 2 3; do for v in base onesolve; do run "$v rep$rep" $V/$v.so; done; done; } | tee gpurun_out/r06_aj_ab.txt

#!/bin/bash
# round 6: scheduler strategies on lcp_quad.hip (run-time sizes: what every ContactWorld gets) - bench.py's general_kernel companion and the world with detection
cd /root/repo; mkdir -p gpurun_out
V=lcp_physics_amd/csrc/variants
{ for rep in 1 2; do for v in q_base q_memcl q_iterilp; do
  LCP_HIP_LIB=$V/$v.so timeout 300 python bench.py --no-cpu-baseline --sustain 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['general_kernel']; print('$v rep$rep: headline %.4g  general_kernel %.4g (%.5f ms)' % (d['value'], g['value'], g['ms_per_step']))"
  LCP_HIP_LIB=$V/$v.so timeout 300 python tools/bench_world.py --cpu-scenes 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v rep$rep: world %.4g solve %.5f ms move %.5f ms' % (d['value'], d['solve_dynamics_ms'], d['move_find_contacts_ms']))"
done; done; } | tee gpurun_out/r06_ak_ab.txt

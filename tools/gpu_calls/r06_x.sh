#!/bin/bash
# round 6: (1) recorded 20-body world roll-outs under sync-debug-mode "error" (VERDICT item 7's bench line);
# (2) A/B of the machine scheduler's strategy on the headline unit (lcp_quad_n15e3.hip), same box, interleaved
cd /root/repo; mkdir -p gpurun_out
timeout 600 python tools/bench_world.py --batch 1024 --nbox 19 --maxc 48 --steps 20 --settle 10 --record 4 --cpu-scenes 1 2>gpurun_out/r06_x_world.err | tail -1 > gpurun_out/r06_x_world20.json
cut -c1-1500 gpurun_out/r06_x_world20.json; tail -3 gpurun_out/r06_x_world.err
V=lcp_physics_amd/csrc/variants
for rep in 1 2; do for v in base maxilp iterilp memcl; do
  LCP_HIP_LIB=$V/$v.so timeout 200 python bench.py --no-cpu-baseline --no-companions --sustain 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v rep $rep: value %.4g ms/step %.5f fwd %.5f bwd %.5f' % (d['value'], d['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"
done; done | tee gpurun_out/r06_x_ab_sched.txt

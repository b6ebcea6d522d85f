#!/bin/bash
# round 6 A/B: the dense-boundary backward of configs[4] with / without the iterate loads hoisted into the prologue (same box, interleaved)
cd /root/repo; mkdir -p gpurun_out
V=lcp_physics_amd/csrc/variants
for rep in 1 2; do for v in hoist nohoist; do
  LCP_HIP_LIB=$V/$v.so timeout 200 python bench.py --config 4 --mode dense --no-cpu-baseline --no-companions --sustain 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v rep $rep: value %.4g ms/step %.4f fwd %.4f bwd %.4f' % (d['value'], d['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"
done; done | tee gpurun_out/r06_w_ab.txt
timeout 300 python bench.py --nbox 19 --pts 2 --batch 1024 --bwd physical --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_w_bench_20bodies.json
cut -c1-300 gpurun_out/r06_w_bench_20bodies.json

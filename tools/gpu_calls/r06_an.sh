#!/bin/bash
# round 6: the 64-row instantiation of the body-space step kernel (18 .. 20 bodies): the tests that run those sizes, then the 20-body lines
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_dense_step.py tests/test_hip_primal.py tests/test_hip_step_backward.py tests/test_hip_reentrancy.py -m gpu -q -s 2>&1 | grep -v "RuntimeWarning\|v_new = fn\|dp = fn\|^$" | tail -30
timeout 300 python bench.py --nbox 19 --pts 2 --batch 1024 --bwd physical --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_an_step20.json; python -c "
import json; d=json.loads(open('gpurun_out/r06_an_step20.json').read()); r=d['roofline']; print('20 bodies step: %.4g steps/s fwd %.4f ms bwd %.4f ms' % (d['value'], r['fwd_ms'], r['bwd_ms']))"
timeout 600 python tools/bench_world.py --batch 1024 --nbox 19 --maxc 48 --steps 20 --settle 10 --record 4 --cpu-scenes 0 2>/dev/null | tail -1 | cut -c1-900

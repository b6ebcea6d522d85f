#!/bin/bash
# round 6: the 64-row instantiation of the post-stabilisation kernel - its tests, then the 20-body world with post-stabilisation
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_dense_step.py tests/test_hip_primal.py tests/test_hip_contacts.py -m gpu -q -s 2>&1 | grep "post-stabilisation, worst\|passed\|failed\|Error" | cut -c1-500
timeout 600 python tools/bench_world.py --batch 1024 --nbox 19 --maxc 48 --steps 20 --settle 10 --record 4 --cpu-scenes 0 --post-stab 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('20-body world, post-stab: %.4g steps/s, recorded %.4g' % (d['value'], d['recorded']['value']))"

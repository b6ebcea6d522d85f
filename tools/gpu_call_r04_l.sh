#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r04_gputests.log 2>&1; echo "suite rc=$?"; tail -4 $O/r04_gputests.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('headline %.2f M ms %.4f | general_kernel %.2f M (%.4f ms) bitwise %s | with_multipliers %.2f M | graph %.4f' % (j['value']/1e6, j['ms_per_step'], j['general_kernel']['value']/1e6, j['general_kernel']['ms_per_step'], j['general_kernel']['new_v_bitwise_equal_to_the_timed_kernel'], j['with_multipliers']['value']/1e6, j['graph_ms_per_step']))"
timeout 300 python tools/bench_world.py --cpu-scenes 0 2>/dev/null | tail -1 | cut -c1-300
timeout 300 python tools/bench_world.py --cpu-scenes 0 --graph 2>/dev/null | tail -1 | cut -c1-200

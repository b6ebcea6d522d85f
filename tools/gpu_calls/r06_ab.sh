#!/bin/bash
# round 6: why is the TIMED region of the 32768-scene lines slower than the sustained loop and the graph replay of the same run (collect_round r06)?
cd /root/repo; mkdir -p gpurun_out
run() { local n=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-companions 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$n: value %.4g ms/step %.5f fwd %.5f bwd %.5f sustained %.4g graph %s' % (d['value'], d['ms_per_step'], r['fwd_ms'], r['bwd_ms'], (d.get('sustained') or {}).get('value', 0), d.get('graph_ms_per_step')))"
}
{
run "config3 default" --config 3
run "config3 default again" --config 3
run "config3 spinup 2 s" --config 3 --spinup 2
run "config3 warmup 500" --config 3 --warmup 500
run "config3 steps 1000" --config 3 --steps 1000
run "config4 default" --config 4
run "config4 spinup 2 s" --config 4 --spinup 2
} | tee gpurun_out/r06_ab_timed_region.txt

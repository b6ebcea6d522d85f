#!/bin/bash
# round 6: contact record loads hoisted in assemble_q (quad) and lcp_fwd_solo (+ LDS staging of the gathers): parity, then the bench lines
cd /root/repo; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_headline_parity.py tests/test_hip_step_backward.py tests/test_hip_contacts.py tests/test_hip_solo.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r06_z_tests.txt
cat gpurun_out/r06_z_tests.txt
run() { local n=$1; shift
  timeout 300 python bench.py "$@" --no-cpu-baseline --no-companions --sustain 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$n: value %.4g ms/step %.5f fwd %.5f bwd %.5f' % (d['value'], d['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"
}
{ for rep in 1 2; do run headline; run config1 --config 1; run "bwd physical" --bwd physical; done; } | tee gpurun_out/r06_z_bench.txt

#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_primal.py tests/test_hip_step_backward.py -q -x > $O/r04_bsweep_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r04_bsweep_tests.log | cut -c1-300
timeout 600 python -m pytest tests/test_hip_headline_parity.py -q -x -s -k "configs4" > $O/r04_bsweep_headline.log 2>&1; echo "headline rc=$?"; tail -5 $O/r04_bsweep_headline.log | cut -c1-400
timeout 300 python bench.py --config 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('config4 %.2f M ms %.4f | fwd_ms %s bwd_ms %s | parity iters_equal %s' % (j['value']/1e6, j['ms_per_step'], r.get('kernel_ms_forward', r.get('forward_ms')), r.get('kernel_ms_backward', r.get('backward_ms')), j.get('parity',{}).get('iters_equal_frac')))
print({k: r[k] for k in r if 'ms' in k or 'frac' in k})"
LCP_HIP_LIB=$PWD/tools/liblcp_primalprof.so timeout 200 python tools/config5_phases.py 4096 > $O/r04_bsweep_phases.txt 2>&1; tail -14 $O/r04_bsweep_phases.txt | cut -c1-200

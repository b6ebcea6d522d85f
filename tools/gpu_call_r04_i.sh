#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_headline_parity.py -q -m gpu -x -k "dense or mixed or stack_scenes or fixture" > $O/r04_dense_tests.log 2>&1; echo "dense tests rc=$?"; tail -3 $O/r04_dense_tests.log
for rep in 1 2; do timeout 300 python bench.py --mode dense --no-cpu-baseline --no-companions 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('dense: %.3f M fwd+bwd  ms/step %.4f  fwd %.4f ms  bwd %.4f ms' % (j['value']/1e6, j['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_dense -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode dense --steps 20 --warmup 5 --no-cpu-baseline --no-companions --spinup 0 > $GRAFT_REPO_ROOT/$O/prof_dense.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_dense -name "*.db" | head -1); python tools/rocprof_summary.py $f > $O/r04_dense_kernel_stats.txt; rm -rf $O/prof_dense
head -14 $O/r04_dense_kernel_stats.txt | cut -c1-200

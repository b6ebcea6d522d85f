#!/bin/bash
# round 6, call l: kernel trace of the split backward
O=$PWD/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/prof_split -o split -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-companions --sustain 0 > $O/prof_split.log 2>&1
python $R/tools/rocprof_summary.py $O/prof_split 2>/dev/null | head -20 || ls -R $O/prof_split | head

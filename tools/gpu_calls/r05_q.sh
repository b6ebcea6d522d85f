#!/bin/bash
# round 5, call q: where does the world with post-stabilisation spend its step ?  (kernel trace of tools/bench_world.py --post-stab)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r05_world_ps -o trace -- python $ROOT/tools/bench_world.py --cpu-scenes 0 --steps 50 --post-stab > $OUT/prof_r05_world_ps.log 2>&1
cd $ROOT
f=$(find $OUT/prof_r05_world_ps -name "*.db" | head -1)
python tools/rocprof_summary.py $f > $OUT/r05_world_post_stab_kernel_stats.txt
rm -rf $OUT/prof_r05_world_ps/
head -14 $OUT/r05_world_post_stab_kernel_stats.txt | cut -c1-190

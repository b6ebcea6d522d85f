#!/bin/bash
# End-of-round-2 measurements of the jointed / differentiable paths on the GPU box: the chain world (body-space kernel with 24
# equality rows against the generic kernel), the batched mass-inference experiment, and rocprofv3 kernel traces of both plus
# BASELINE config 5.  Summaries land in gpurun_out/; copy them into profiles/ by hand.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python tools/experiments/chain_world.py 2>/dev/null | tail -1 > $OUT/r02_bench_chain_world.json
python tools/experiments/chain_world.py generic 2>/dev/null | tail -1 >> $OUT/r02_bench_chain_world.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_chain -o trace -- python $ROOT/tools/experiments/chain_world.py > $OUT/prof_chain.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_inf -o trace -- python $ROOT/tools/experiments/mass_inference.py --batch 4096 --iters 3 > $OUT/prof_inf.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_c5 -o trace -- python $ROOT/tools/config5_phases.py 4096 > $OUT/prof_c5.log 2>&1
cd $ROOT
for t in chain inf c5; do
  f=$(find $OUT/prof_$t -name "*.db" | head -1)
  python tools/rocprof_summary.py $f > $OUT/r02_${t}_kernel_stats.txt
  rm -rf $OUT/prof_$t
done
head -12 $OUT/r02_inf_kernel_stats.txt | cut -c1-160

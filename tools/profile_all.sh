#!/bin/bash
# Collect the round's profiles on the GPU box: kernel trace + stats, then PMC counters in separate passes
# (never combined with tracing domains).  Outputs land in gpurun_out/prof_<tag>/ ; summaries are made by
# tools/rocprof_summary.py and tools/pmc_summary.py and copied into profiles/ by hand.
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-companions --spinup 0 --event-samples 8 ${EXTRA}"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_trace -o trace -- $BENCH > $OUT/prof_${TAG}_trace.log 2>&1
echo "trace rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $ctr -d $OUT/prof_${TAG}_pmc_$name -o pmc -- $BENCH > $OUT/prof_${TAG}_pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
ls $OUT | grep prof_${TAG}
# the world with contact detection (kernel trace only)
[ -n "$EXTRA" ] || timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_world -o trace -- python $ROOT/tools/bench_world.py --cpu-scenes 0 --steps 50 > $OUT/prof_${TAG}_world.log 2>&1
echo "world trace rc=$?"
cd $ROOT
for d in $OUT/prof_${TAG}_trace $OUT/prof_${TAG}_world; do
  f=$(find $d -name "*.db" | head -1)
  python tools/rocprof_summary.py $f > $d.summary.txt
done
python tools/pmc_summary.py $OUT/prof_${TAG}_pmc_* > $OUT/prof_${TAG}_pmc.summary.txt
rm -rf $OUT/prof_${TAG}_*/      # the raw databases are large; the summaries are what gets committed

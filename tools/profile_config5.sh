#!/bin/bash
# BASELINE config 5 (4096 scenes x 64 contacts) on the GPU box: rocprofv3 kernel trace + stats, PMC counters in their own
# passes (never combined with tracing domains), and - when tools/liblcp_primalprof.so exists - the in-kernel phase profile.
# Summaries land in gpurun_out/prof_<tag>_config5.*; copy them into profiles/ by hand.
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --config 4 --batch ${BATCH:-4096} --steps 10 --warmup 2 --no-cpu-baseline --no-companions --spinup 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_c5_trace -o trace -- $CMD > $OUT/prof_${TAG}_c5_trace.log 2>&1
echo "trace rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  name=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $ctr -d $OUT/prof_${TAG}_c5_pmc_$name -o pmc -- $CMD > $OUT/prof_${TAG}_c5_pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
cd $ROOT
f=$(find $OUT/prof_${TAG}_c5_trace -name "*.db" | head -1)
python tools/rocprof_summary.py $f > $OUT/prof_${TAG}_config5_kernel_stats.txt
python tools/pmc_summary.py $OUT/prof_${TAG}_c5_pmc_* > $OUT/prof_${TAG}_config5_pmc.txt
rm -rf $OUT/prof_${TAG}_c5_*/
if [ -f tools/liblcp_primalprof.so ]; then      # `make -C lcp_physics_amd/csrc primalprof`: lcp_primal.hip with -DLCP_PRIMAL_PROFILE, cycles per phase
  LCP_HIP_LIB=$ROOT/tools/liblcp_primalprof.so timeout 200 python tools/config5_phases.py ${BATCH:-4096} > $OUT/prof_${TAG}_config5_phases.txt 2>&1
fi
timeout 200 python tools/config5_phases.py ${BATCH:-4096} big > $OUT/prof_${TAG}_config5_contact_space.txt 2>&1
tail -4 $OUT/prof_${TAG}_config5_phases.txt
head -6 $OUT/prof_${TAG}_config5_kernel_stats.txt

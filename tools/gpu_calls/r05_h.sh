#!/bin/bash
# round 5, call h: the dense boundary at 17..64 contacts after classify + extract (one pass over the dense tensors), the pinned form behind it, 16-byte gradient stores
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_headline_parity.py tests/test_hip_parity.py tests/test_hip_primal.py -q -m gpu -k "pile_dense or config5 or dense_boundary or big_kernel" 2>&1 | grep -E "passed|failed|^FAILED|Error" | cut -c1-300
run() { name=$1; shift; timeout 300 "$@" > $O/r05_$name.json 2> $O/r05_$name.err; tail -1 $O/r05_$name.json | cut -c1-200; }
run h_bench_config5_dense python bench.py --config 4 --mode dense --no-cpu-baseline
run h_bench_config5 python bench.py --config 4 --no-cpu-baseline
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
tag=dense5
BENCH="python $ROOT/bench.py --config 4 --mode dense --steps 10 --warmup 2 --no-cpu-baseline --no-companions --spinup 0 --event-samples 8"
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$O/prof_r05_${tag}_trace -o trace -- $BENCH > $ROOT/$O/prof_r05_${tag}_trace.log 2>&1; echo "trace $tag rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr -d $ROOT/$O/prof_r05_${tag}_pmc_$ctr -o pmc -- $BENCH > $ROOT/$O/prof_r05_${tag}_pmc_$ctr.log 2>&1; echo "pmc $tag $ctr rc=$?"
done
cd $ROOT
f=$(find $O/prof_r05_${tag}_trace -name "*.db" | head -1)
python tools/rocprof_summary.py $f > $O/r05_h_${tag}_kernel_stats.txt
python tools/pmc_summary.py $O/prof_r05_${tag}_pmc_* > $O/r05_h_pmc_${tag}.txt
head -9 $O/r05_h_${tag}_kernel_stats.txt | cut -c1-150
grep -E "FETCH_SIZE|WRITE_SIZE" $O/r05_h_pmc_${tag}.txt | head -8 | cut -c1-190
rm -rf $O/prof_r05_*/

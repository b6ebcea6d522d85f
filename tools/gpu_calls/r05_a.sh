#!/bin/bash
# round 5, call a: the dense boundary at configs[4] (tests, bench lines, trace + traffic), kernel outputs of the headline cases for the
# offline oracle report (tools/experiments/headline_dump.py), the headline line at the start of the round
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_headline_parity.py -q -m gpu -k "pile_dense" -s > $O/r05_a_tests.log 2>&1; echo "tests rc=$?"; grep -E "^FAILED|passed|failed|headline parity" $O/r05_a_tests.log | cut -c1-1500
timeout 600 python tools/experiments/headline_dump.py dump $O/r05_dump > $O/r05_a_dump.log 2>&1; echo "dump rc=$?"; tail -3 $O/r05_a_dump.log | cut -c1-300
run() { name=$1; shift; timeout 300 "$@" > $O/r05_$name.json 2> $O/r05_$name.err; tail -1 $O/r05_$name.json | cut -c1-200; }
run a_bench_fused python bench.py
run a_bench_config5_dense python bench.py --config 4 --mode dense --cpu-budget 5
run a_bench_config5_dense_cs python bench.py --config 4 --mode dense --contact-space --cpu-budget 5
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for v in "dense5:" "dense5cs:--contact-space"; do
  tag=${v%%:*}; extra=${v#*:}
  BENCH="python $ROOT/bench.py --config 4 --mode dense $extra --steps 10 --warmup 2 --no-cpu-baseline --no-companions --spinup 0 --event-samples 8"
  timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$O/prof_r05_${tag}_trace -o trace -- $BENCH > $ROOT/$O/prof_r05_${tag}_trace.log 2>&1; echo "trace $tag rc=$?"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr -d $ROOT/$O/prof_r05_${tag}_pmc_$ctr -o pmc -- $BENCH > $ROOT/$O/prof_r05_${tag}_pmc_$ctr.log 2>&1; echo "pmc $tag $ctr rc=$?"
  done
done
cd $ROOT
for tag in dense5 dense5cs; do
  f=$(find $O/prof_r05_${tag}_trace -name "*.db" | head -1)
  python tools/rocprof_summary.py $f > $O/r05_${tag}_kernel_stats.txt
  python tools/pmc_summary.py $O/prof_r05_${tag}_pmc_* > $O/r05_pmc_${tag}.txt
  head -8 $O/r05_${tag}_kernel_stats.txt | cut -c1-150
done
rm -rf $O/prof_r05_*/

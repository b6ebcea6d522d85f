#!/bin/bash
# round 5, call z: the bench lines again at the final HEAD (roofline.traffic / counters now quoted from profiles/r05_*.json; parity with the own-iterate fields)
TAG=r05
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out
mkdir -p $O
cd $ROOT
run() { name=$1; shift; timeout 300 "$@" > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err; tail -1 $O/${TAG}_$name.json | cut -c1-140; }
run bench_fused python bench.py
run bench_driver_form python bench.py --steps 20 --warmup 5
run bench_dense python bench.py --mode dense --cpu-budget 3
run bench_dense_contact_space python bench.py --mode dense --contact-space --cpu-budget 3
run bench_fused_physical_bwd python bench.py --bwd physical --no-cpu-baseline
run bench_config2_fwd_only python bench.py --config 1 --no-cpu-baseline
run bench_config5 python bench.py --config 4
run bench_config5_dense python bench.py --config 4 --mode dense --cpu-budget 5
run bench_config5_dense_contact_space python bench.py --config 4 --mode dense --contact-space --no-cpu-baseline

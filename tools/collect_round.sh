#!/bin/bash
# One GPU call that refreshes every measured artefact of a round: bench lines of all BASELINE configs, the worlds with contact
# detection, config 5 through both boundaries, rocprofv3 kernel trace + PMC passes (profile_all.sh / profile_config5.sh).
# Outputs land in gpurun_out/<tag>_*; copy the ones to keep into profiles/.
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out
mkdir -p $O
cd $ROOT
run() { name=$1; shift; timeout 300 "$@" > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err; tail -1 $O/${TAG}_$name.json | cut -c1-200; }
run bench_fused python bench.py
run bench_dense python bench.py --mode dense --cpu-budget 3
run bench_dense_contact_space python bench.py --mode dense --contact-space --cpu-budget 3
run bench_driver_form python bench.py --steps 20 --warmup 5
run bench_fused_physical_bwd python bench.py --bwd physical --no-cpu-baseline
run bench_config2_fwd_only python bench.py --config 1 --no-cpu-baseline
run bench_config4_on_1gpu python bench.py --batch 32768 --no-cpu-baseline
run bench_fused_8contacts python bench.py --pts 2 --no-cpu-baseline
run bench_config5 python bench.py --config 4
run bench_midsize_24 python tools/bench_midsize.py 6 4
run bench_midsize_32 python tools/bench_midsize.py 8 4
run bench_world python tools/bench_world.py --cpu-scenes 2
run bench_world_graph python tools/bench_world.py --cpu-scenes 0 --graph
run bench_world_11bodies python tools/bench_world.py --nbox 10 --box 24 --maxc 32 --cpu-scenes 0
run bench_world_6bodies python tools/bench_world.py --nbox 5 --box 40 --cpu-scenes 0
run batch_curve_2box python tools/bench_batch_curve.py 2
{ timeout 300 python tools/experiments/grad_demo_rollout.py --rep 128 --eager --count; timeout 300 python tools/experiments/grad_demo_rollout.py --rep 128; timeout 300 python tools/experiments/grad_demo_rollout.py --rep 512 --eager; timeout 300 python tools/experiments/grad_demo_rollout.py --rep 512; } 2>/dev/null | grep "^{" > $O/${TAG}_grad_demo_rollout.json
{ timeout 300 python tools/experiments/mass_inference.py --batch 4096 --count; timeout 300 python tools/experiments/mass_inference.py --batch 4096 --graph; } 2>/dev/null | grep "^{" > $O/${TAG}_mass_inference.json
run batch_curve_4box python tools/bench_batch_curve.py 4
EXTRA="" bash tools/profile_all.sh $TAG > $O/${TAG}_profile_all.log 2>&1
EXTRA="--mode dense" bash tools/profile_all.sh ${TAG}dense > $O/${TAG}_profile_dense.log 2>&1
bash tools/profile_config5.sh $TAG > $O/${TAG}_profile_config5.log 2>&1
ls $O | grep "^prof_${TAG}\|^${TAG}_" | head -60
# in-kernel phase profile of the headline forward (needs `make -C lcp_physics_amd/csrc quadprof`)
if [ -f tools/liblcp_quadprof.so ]; then
  LCP_HIP_LIB=$ROOT/tools/liblcp_quadprof.so timeout 200 python tools/gpu_phase_profile_quad.py 4096 4 > $O/${TAG}_quad_phase_profile.txt 2>&1
  LCP_HIP_LIB=$ROOT/tools/liblcp_quadprof.so timeout 200 python tools/gpu_phase_profile_quad.py 32768 4 >> $O/${TAG}_quad_phase_profile.txt 2>&1
fi

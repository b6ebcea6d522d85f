#!/bin/bash
# One GPU call that refreshes every measured artefact of a round: bench lines of all BASELINE configs through both boundaries, the worlds
# with contact detection, rocprofv3 kernel trace + PMC passes (profile_all.sh / profile_config5.sh / the dense configs[4] boundary), the
# in-kernel phase profiles, the whole-batch parity of configs[3].  Outputs land in gpurun_out/<tag>_*; copy the ones to keep into profiles/.
# Before the call, in the build container:  make -C lcp_physics_amd/csrc quadprof primalprof soloprof ; python tools/kernel_resources.py > profiles/<tag>_kernel_resources.json
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=$ROOT/gpurun_out
mkdir -p $O
cd $ROOT
# PROFILES_ONLY=1: skip the bench / world / experiment lines and collect only the rocprofv3 traces, PMC passes and phase profiles
run() { [ -n "$PROFILES_ONLY" ] && return 0; name=$1; shift; timeout 300 "$@" > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err; tail -1 $O/${TAG}_$name.json | cut -c1-160; }
run bench_fused python bench.py
run bench_driver_form python bench.py --steps 20 --warmup 5
run bench_dense python bench.py --mode dense --cpu-budget 3
run bench_dense_contact_space python bench.py --mode dense --contact-space --cpu-budget 3
run bench_fused_physical_bwd python bench.py --bwd physical --no-cpu-baseline
run bench_config2_fwd_only python bench.py --config 1 --no-cpu-baseline
run bench_config4_on_1gpu python bench.py --batch 32768 --no-cpu-baseline
run bench_fused_8contacts python bench.py --pts 2 --no-cpu-baseline
run bench_config5 python bench.py --config 4
run bench_config1_parity python bench.py --config 1
run bench_config2_parity python bench.py --config 2
run bench_config3_parity python bench.py --config 3
run bench_config4_parity python bench.py --config 4 --no-cpu-baseline
run bench_config5_dense python bench.py --config 4 --mode dense --cpu-budget 5
run bench_config5_dense_contact_space python bench.py --config 4 --mode dense --contact-space --no-cpu-baseline
run bench_2ranks_one_device python bench.py --gpus 2 --share-devices --steps 20 --warmup 5 --no-cpu-baseline
# BENCH_ONLY=1: just the bench.py lines above (e.g. after tools/make_profile_json.py has refreshed the JSONs they quote)
[ -n "$BENCH_ONLY" ] && exit 0
run bench_midsize_24 python tools/bench_midsize.py 6 4
run bench_midsize_32 python tools/bench_midsize.py 8 4
run bench_world python tools/bench_world.py --cpu-scenes 2
run bench_world_graph python tools/bench_world.py --cpu-scenes 0 --graph
run bench_world_post_stab python tools/bench_world.py --cpu-scenes 0 --post-stab
run bench_world_11bodies python tools/bench_world.py --nbox 10 --box 24 --maxc 32 --cpu-scenes 0
run bench_world_6bodies python tools/bench_world.py --nbox 5 --box 40 --cpu-scenes 0
# (round 6) 20 bodies: the generic kernels forward, and RECORDED roll-outs with their backward under torch's sync-debug-mode "error"
run bench_world_20bodies python tools/bench_world.py --batch 1024 --nbox 19 --maxc 48 --steps 20 --settle 10 --record 4 --cpu-scenes 1
run bench_world_20bodies_post_stab python tools/bench_world.py --batch 1024 --nbox 19 --maxc 48 --steps 20 --settle 10 --record 4 --cpu-scenes 0 --post-stab
run bench_step_20bodies_physical python bench.py --nbox 19 --pts 2 --batch 1024 --bwd physical --no-cpu-baseline
[ -n "$PROFILES_ONLY" ] || timeout 900 python bench.py --gpus 8 --share-devices --no-cpu-baseline > $O/${TAG}_bench_8ranks_one_device.json 2> $O/${TAG}_bench_8ranks_one_device.err
run batch_curve_2box python tools/bench_batch_curve.py 2
run batch_curve_4box python tools/bench_batch_curve.py 4
run engine_latency python tools/experiments/engine_latency.py
if [ -z "$PROFILES_ONLY" ]; then
timeout 300 python tools/experiments/poststab_time.py > $O/${TAG}_poststab_time.txt 2>/dev/null
{ timeout 300 python tools/experiments/grad_demo_rollout.py --rep 128 --eager --count; timeout 300 python tools/experiments/grad_demo_rollout.py --rep 128; timeout 300 python tools/experiments/grad_demo_rollout.py --rep 512 --eager; timeout 300 python tools/experiments/grad_demo_rollout.py --rep 512; } 2>/dev/null | grep "^{" > $O/${TAG}_grad_demo_rollout.json
{ timeout 300 python tools/experiments/mass_inference.py --batch 4096 --count; timeout 300 python tools/experiments/mass_inference.py --batch 4096 --graph; } 2>/dev/null | grep "^{" > $O/${TAG}_mass_inference.json
timeout 600 python tools/experiments/config3_all_shards_parity.py > $O/${TAG}_config3_all_shards_parity.json 2> $O/${TAG}_config3_all_shards_parity.err; tail -c 400 $O/${TAG}_config3_all_shards_parity.json
fi
EXTRA="" bash tools/profile_all.sh $TAG > $O/${TAG}_profile_all.log 2>&1
EXTRA="--mode dense" bash tools/profile_all.sh ${TAG}dense > $O/${TAG}_profile_dense.log 2>&1
bash tools/profile_config5.sh $TAG > $O/${TAG}_profile_config5.log 2>&1
# the dense boundary at configs[4], body space and contact space: trace + traffic + issue counters (MFMA for the contact-space LU)
cd /tmp && export TMPDIR=/tmp
for v in "dense5:" "dense5cs:--contact-space"; do
  tag=${v%%:*}; extra=${v#*:}
  BENCH="python $ROOT/bench.py --config 4 --mode dense $extra --steps 10 --warmup 2 --no-cpu-baseline --no-companions --spinup 0 --event-samples 8"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_${tag}_trace -o trace -- $BENCH > $O/prof_${TAG}_${tag}_trace.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"; do
    name=$(echo $ctr | tr ' ' '_' | cut -c1-24)
    timeout 300 rocprofv3 --pmc $ctr -d $O/prof_${TAG}_${tag}_pmc_$name -o pmc -- $BENCH > $O/prof_${TAG}_${tag}_pmc_$name.log 2>&1
  done
done
cd $ROOT
for tag in dense5 dense5cs; do
  f=$(find $O/prof_${TAG}_${tag}_trace -name "*.db" | head -1)
  python tools/rocprof_summary.py $f > $O/${TAG}_${tag}_kernel_stats.txt
  python tools/pmc_summary.py $O/prof_${TAG}_${tag}_pmc_* > $O/${TAG}_pmc_${tag}.txt
done
rm -rf $O/prof_${TAG}_dense5*/
ls $O | grep "^prof_${TAG}\|^${TAG}_" | wc -l
# in-kernel phase profile of the headline forward (needs `make -C lcp_physics_amd/csrc quadprof`)
if [ -f tools/liblcp_quadprof.so ]; then
  LCP_HIP_LIB=$ROOT/tools/liblcp_quadprof.so timeout 200 python tools/gpu_phase_profile_quad.py 4096 4 > $O/${TAG}_quad_phase_profile.txt 2>&1
  LCP_HIP_LIB=$ROOT/tools/liblcp_quadprof.so timeout 200 python tools/gpu_phase_profile_quad.py 32768 4 >> $O/${TAG}_quad_phase_profile.txt 2>&1
fi
if [ -f tools/liblcp_soloprof.so ]; then
  LCP_HIP_LIB=$ROOT/tools/liblcp_soloprof.so timeout 200 python tools/gpu_phase_profile_solo.py > $O/${TAG}_solo_phase_profile.txt 2>&1
fi
timeout 300 python -c "
import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1; tail -1 $O/${TAG}_smoke.txt

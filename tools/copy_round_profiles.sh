#!/bin/bash
# After `bash tools/collect_round.sh <tag>` (on the GPU box; its outputs merge back into gpurun_out/): copy the summaries to keep into profiles/
# under the names bench.py, DESIGN.md and profiles/README.md use, and make the stamped JSONs bench.py quotes.
#   bash tools/copy_round_profiles.sh r06 [--lines-only]
TAG=${1:-r06}
cd "$(dirname "$0")/.."
O=gpurun_out; P=profiles
LINES="bench_fused bench_driver_form bench_dense bench_dense_contact_space bench_fused_physical_bwd bench_config2_fwd_only bench_config4_on_1gpu bench_fused_8contacts bench_config5 bench_config5_dense bench_config5_dense_contact_space bench_config1_parity bench_config2_parity bench_config3_parity bench_config4_parity bench_2ranks_one_device"
for f in $LINES; do [ -s $O/${TAG}_$f.json ] && cp $O/${TAG}_$f.json $P/; done
[ "$2" = "--lines-only" ] && exit 0
cp $O/prof_${TAG}_trace.summary.txt $P/${TAG}_kernel_stats.txt
cp $O/prof_${TAG}_pmc.summary.txt $P/${TAG}_pmc.txt
cp $O/prof_${TAG}dense_trace.summary.txt $P/${TAG}_dense_kernel_stats.txt
cp $O/prof_${TAG}dense_pmc.summary.txt $P/${TAG}_pmc_dense.txt
cp $O/prof_${TAG}_world.summary.txt $P/${TAG}_world_kernel_stats.txt
cp $O/prof_${TAG}_config5_kernel_stats.txt $P/${TAG}_config5_primal_kernel_stats.txt
cp $O/prof_${TAG}_config5_pmc.txt $P/${TAG}_config5_primal_pmc.txt
cp $O/prof_${TAG}_config5_phases.txt $P/${TAG}_config5_primal_phases.txt
for f in dense5_kernel_stats.txt dense5cs_kernel_stats.txt pmc_dense5.txt pmc_dense5cs.txt quad_phase_profile.txt solo_phase_profile.txt poststab_time.txt smoke.txt; do cp $O/${TAG}_$f $P/; done
for f in batch_curve_2box batch_curve_4box bench_midsize_24 bench_midsize_32 bench_world bench_world_graph bench_world_post_stab bench_world_11bodies bench_world_6bodies bench_world_20bodies bench_world_20bodies_post_stab bench_step_20bodies_physical bench_8ranks_one_device engine_latency grad_demo_rollout mass_inference config3_all_shards_parity; do [ -s $O/${TAG}_$f.json ] && cp $O/${TAG}_$f.json $P/; done
python tools/make_profile_json.py $P/${TAG}_pmc.txt $P/${TAG}_pmc_dense.txt $P/${TAG}_config5_primal_pmc.txt $P/${TAG}_pmc_dense5.txt $P/${TAG}_pmc_dense5cs.txt $TAG > /dev/null
python tools/kernel_resources.py > $P/${TAG}_kernel_resources.json
echo "profiles/${TAG}_* refreshed; now: BENCH_ONLY=1 bash tools/collect_round.sh $TAG (on the GPU), then bash tools/copy_round_profiles.sh $TAG --lines-only"

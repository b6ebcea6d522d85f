#!/bin/bash
# runs ON the GPU box, from tools/stage_reference.sh (measurement only; needs the staged reference).  TAG names the outputs (default r05).
TAG=${TAG:-r05}
O=gpurun_out; mkdir -p $O
export LCP_REFERENCE_ROOT=$PWD/oracle/_ref/stage PYTHONDONTWRITEBYTECODE=1
nproc; python -c "import torch; print(torch.get_num_threads())"
timeout 900 python tools/experiments/reference_world_plugin.py > $O/${TAG}_reference_world_plugin.json 2> $O/${TAG}_reference_world_plugin.err; echo "plugin rc=$?"; tail -c 300 $O/${TAG}_reference_world_plugin.err
python - <<PY
import json
try:
    j = json.load(open("gpurun_out/${TAG}_reference_world_plugin.json"))
    for k, v in j["scenes"].items():
        print("%-18s steps %3d  max|dp| %.2e (fused %.2e)  counts equal %d  ms/step ref %.2f hip %.2f fused %.2f (first step ref %.1f hip %.1f)" % (k, v["steps"], v["max_abs_pose_diff"], v["max_abs_pose_diff_fused_engine"], v["contact_counts_equal_steps"], v["ms_per_step_reference_engine"], v["ms_per_step_hip_engine"], v["ms_per_step_hip_fused_engine"], v["ms_first_step_reference_engine"], v["ms_first_step_hip_engine"]))
    print(j["rollout_gradient"])
except Exception as ex:
    print("ERR", ex)
PY
[ -n "$ONLY_PLUGIN" ] && exit 0                  # (the world comparison alone; the reference timings below take a few minutes)
M="GPU box host ($(nproc) cores, same box as the bench)"
# thread sweep of the unmodified reference on the headline workload: the best thread count is what bench.py quotes
if [ -z "$SKIP_SWEEP" ]; then                    # (SKIP_SWEEP=1: keep the committed sweep, refresh the world comparison and the in-run line only)
timeout 900 python oracle/time_reference.py --batch 1024 --dtype float64 --reps 1 --threads 1,4,8,16,32,64,128 --machine "$M" > $O/${TAG}_reference_cpu_timing_f64_sweep.jsonl 2>/dev/null; cut -c1-20,180-330 $O/${TAG}_reference_cpu_timing_f64_sweep.jsonl
timeout 600 python oracle/time_reference.py --batch 1024 --dtype float32 --reps 1 --threads 8,16 --machine "$M" > $O/${TAG}_reference_cpu_timing_f32_sweep.jsonl 2>/dev/null; cut -c1-20,180-330 $O/${TAG}_reference_cpu_timing_f32_sweep.jsonl
timeout 600 python oracle/time_reference.py --batch 128 --dtype float64 --reps 1 --pile --machine "$M" > $O/${TAG}_reference_cpu_timing_pile.jsonl 2>/dev/null; cut -c1-20,180-330 $O/${TAG}_reference_cpu_timing_pile.jsonl
fi
# the bench line WITH the reference timed inside the run (cpu_reference.measured_in_this_run = true)
timeout 600 python bench.py > $O/${TAG}_bench_fused_with_reference.json 2> $O/${TAG}_bench_fused_with_reference.err; python -c "
import json; j = json.loads(open('$O/${TAG}_bench_fused_with_reference.json').read().strip().splitlines()[-1]); print(j['value'], j['cpu_reference'])"

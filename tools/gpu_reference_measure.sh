#!/bin/bash
# runs ON the GPU box, from tools/stage_reference.sh (measurement only; needs the staged reference)
O=gpurun_out; mkdir -p $O
export LCP_REFERENCE_ROOT=$PWD/oracle/_ref/stage PYTHONDONTWRITEBYTECODE=1
nproc; python -c "import torch; print(torch.get_num_threads())"
timeout 900 python tools/experiments/reference_world_plugin.py > $O/r04_reference_world_plugin.json 2> $O/r04_reference_world_plugin.err; echo "plugin rc=$?"; tail -c 300 $O/r04_reference_world_plugin.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r04_reference_world_plugin.json"))
    for k, v in j["scenes"].items():
        print("%-18s steps %3d  max|dp| %.2e (fused %.2e)  counts equal %d  ms/step ref %.1f hip %.1f" % (k, v["steps"], v["max_abs_pose_diff"], v["max_abs_pose_diff_fused_engine"], v["contact_counts_equal_steps"], v["ms_per_step_reference_engine"], v["ms_per_step_hip_engine"]))
    print(j["rollout_gradient"])
except Exception as ex:
    print("ERR", ex)
PY
[ -n "$ONLY_PLUGIN" ] && exit 0                  # (the world comparison alone; the reference timings below take ~4 minutes)
M="GPU box host ($(nproc) cores, same box as the bench)"
timeout 600 python oracle/time_reference.py --batch 4096 --dtype float64 --reps 1 --machine "$M" > $O/r04_reference_cpu_timing_f64.json 2>/dev/null; cat $O/r04_reference_cpu_timing_f64.json
timeout 600 python oracle/time_reference.py --batch 4096 --dtype float32 --reps 1 --machine "$M" > $O/r04_reference_cpu_timing_f32.json 2>/dev/null; cat $O/r04_reference_cpu_timing_f32.json
timeout 600 python oracle/time_reference.py --batch 256 --dtype float64 --reps 1 --pile --machine "$M" > $O/r04_reference_cpu_timing_pile.json 2>/dev/null; cat $O/r04_reference_cpu_timing_pile.json

#!/bin/bash
O=gpurun_out; mkdir -p $O
LCP_HIP_LIB=$PWD/tools/liblcp_quadprof.so python tools/gpu_phase_profile_quad.py 4096 4 2>&1 | grep -v amdgpu.ids | tee $O/r06_quad_phase_profile.txt
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_step_backward.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_hip_headline_parity.py -m gpu -x -q -k "configs2_4096x16 or configs1_1024x8" 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline > $O/r06_bench_fused_r.json 2>/dev/null; python - $O/r06_bench_fused_r.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); r=d["roofline"]
print("value %.0f" % d["value"], "fwd_ms", r.get("fwd_ms"), "bwd_ms", r.get("bwd_ms"), "frac", r.get("frac"))
PY

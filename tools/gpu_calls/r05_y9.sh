#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for lib in "" lcp_physics_amd/csrc/variants/refine2.so lcp_physics_amd/csrc/variants/floor13.so lcp_physics_amd/csrc/variants/floor11.so; do
  [ -n "$lib" ] && export LCP_HIP_LIB=$PWD/$lib
  timeout 900 python -m pytest tests/test_hip_headline_parity.py -q -m gpu -x -s -k "two_point_shape" 2>&1 | grep "^headline parity" > gpurun_out/r05_two_points_report.txt
  python - "$lib" <<'PY'
import json,sys
s=open("gpurun_out/r05_two_points_report.txt").read()
d=json.loads(s[s.index("{"):])
print(sys.argv[1] or "HEAD", {k:d[k] for k in ("bwd_err_dp_max","bwd_err_phys_max","bwd_err_phys_direct_max","bwd_err_phys_five_max","bwd_own_iterate_err_max","bwd_kkt_resid_all_max","bwd_well_posed_scenes")})
PY
done

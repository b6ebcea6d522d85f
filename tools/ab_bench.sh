#!/bin/bash
# A/B: run bench.py against every library variant under lcp_physics_amd/csrc/variants/
mkdir -p gpurun_out
for lib in lcp_physics_amd/csrc/variants/*.so; do
  n=$(basename $lib .so)
  LCP_HIP_LIB=$PWD/$lib timeout 200 python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline ${EXTRA} > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err
  python - "$n" gpurun_out/ab_$n.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-22s value %10.0f steps/s  fwd %.3f ms  bwd %.3f ms  frac %.4f  status!=0 %d" % (sys.argv[1], d["value"], r["fwd_ms"], r["bwd_ms"], r["frac"], d["config"]["nonzero_status"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done

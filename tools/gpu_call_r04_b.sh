#!/bin/bash
# round 4, second GPU call: the whole GPU suite (dense boundary in body space, contact-space fixture param), bench lines, refine A/B
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x > $O/r04_gputests.log 2>&1; echo "suite rc=$?"; tail -8 $O/r04_gputests.log
timeout 600 python -m pytest tests/test_hip_headline_parity.py -q -m gpu -s 2>&1 | grep "headline parity" | cut -c1-1700 > $O/r04_headline_parity.log; wc -l $O/r04_headline_parity.log
summ() { python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    j = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    r = j["roofline"]; p = j.get("parity") or {}
    print("%-28s %.4g/s ms %.4f fwd %.4f bwd %.4f frac %.3f nec %s graph %s gen %s wm %s | iters %s err_x %s dp %s dQ %s kkt %s phys %s traffic/alg %s" % (
        f, j["value"], j["ms_per_step"], r["fwd_ms"], r["bwd_ms"], r["frac"], r.get("frac_necessary") and round(r["frac_necessary"], 3), j.get("graph_ms_per_step"),
        (j.get("general_kernel") or {}).get("ms_per_step"), (j.get("with_multipliers") or {}).get("ms_per_step"), p.get("iters_delta_hist"),
        p.get("fwd_err_x_max"), p.get("bwd_err_dp_max"), p.get("bwd_err_dQ_max"), p.get("bwd_kkt_resid_max"), p.get("bwd_err_phys_max"), r.get("traffic_over_algorithmic")))
except Exception as ex:
    print(f, "ERR", ex)
PY
}
run() { name=$1; shift; timeout 300 "$@" > $O/$name.json 2> $O/$name.err; summ $name; }
run r04_bench_fused python bench.py
run r04_bench_dense python bench.py --mode dense --cpu-budget 3
run r04_bench_dense_contact_space python bench.py --mode dense --contact-space --cpu-budget 3
run r04_bench_config5 python bench.py --config 4
tail -c 400 $O/r04_bench_config5.err
run r04_bench_driver_form python bench.py --steps 20 --warmup 5
run r04_bench_driver_form_nospin python bench.py --steps 20 --warmup 5 --spinup 0 --no-cpu-baseline
run r04_bench_fused_physical_bwd python bench.py --bwd physical --cpu-budget 3
for v in refine1 refine0; do LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/$v.so run r04_ab_$v python bench.py --cpu-budget 3; done

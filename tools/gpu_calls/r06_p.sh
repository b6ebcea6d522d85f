#!/bin/bash
# round 6, call p: whole GPU suite after the prologue work in lcp_solo / lcp_primal, then the bench lines of configs 1, 2, 4
O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/r06_gputests_p.txt
for c in 1 2 4; do timeout 400 python bench.py --config $c --no-cpu-baseline > $O/r06_bench_config${c}_p.json 2>/dev/null; python - $O/r06_bench_config${c}_p.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); r=d["roofline"]
print(d["config"]["workload"][:40], "value %.0f" % d["value"], "fwd_ms", r.get("fwd_ms"), "bwd_ms", r.get("bwd_ms"), "frac", r.get("frac"))
PY
done
timeout 400 python bench.py --config 4 --mode dense --no-cpu-baseline > $O/r06_bench_config4_dense_p.json 2>/dev/null; tail -1 $O/r06_bench_config4_dense_p.json | cut -c1-120

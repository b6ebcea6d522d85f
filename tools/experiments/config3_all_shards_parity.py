"""One-off (GPU box): the headline parity report of tests/test_hip_headline_parity.py on ALL EIGHT 4096-scene shards of BASELINE configs[3]
(32768 x 16 contacts: the seeds bench.py gives ranks 0 .. 7), every scene of each against the fp64 oracle.
    python tools/experiments/config3_all_shards_parity.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_hip_headline_parity as T

out = {"what": "configs[3] = 8 ranks x 4096 scenes x 16 contacts; fused step + dense backward against the fp64 oracle, every scene", "ranks": {}}
keys = ("scenes", "fwd_err_x_max", "index_set_rows_total", "index_set_mismatches_unmasked", "index_set_mismatches", "index_set_masked_frac",
        "iters_differ_frac", "iters_max_abs_delta", "bwd_err_dp_max", "bwd_well_posed_frac", "bwd_err_dQ_max", "bwd_err_dA_max", "bwd_err_db_max",
        "bwd_kkt_resid_max", "bwd_err_phys_max", "status_nonzero", "bwd_nonfinite_scenes", "bwd_kkt_resid_all_max", "bwd_kkt_resid_all_scenes_over_1e-6", "bwd_input_sensitive_scenes", "bwd_input_sensitivity_max", "fwd_input_sensitivity_err_x_max")
t = time.time()
for rank in range(8):
    K = T.run_kernels("stack", 4096, 4, 1236 + 1000 * rank, "pinned")
    rep, _ = T.report(K, None, input_stability=True)
    out["ranks"][str(rank)] = {k: rep.get(k) for k in keys}
tot = lambda k: sum(r[k] for r in out["ranks"].values())
mx = lambda k: max(r[k] for r in out["ranks"].values())
out["total"] = {"scenes": tot("scenes"), "index_set_rows_total": tot("index_set_rows_total"), "index_set_mismatches_unmasked": tot("index_set_mismatches_unmasked"),
                "index_set_mismatches_on_decisive_rows": tot("index_set_mismatches"), "iters_max_abs_delta": mx("iters_max_abs_delta"),
                "fwd_err_x_max": mx("fwd_err_x_max"), "bwd_err_dp_max": mx("bwd_err_dp_max"), "bwd_err_phys_max": mx("bwd_err_phys_max"),
                "bwd_kkt_resid_max": mx("bwd_kkt_resid_max"), "status_nonzero": tot("status_nonzero"),
                "bwd_nonfinite_scenes (every scene, no filter)": tot("bwd_nonfinite_scenes"), "bwd_kkt_resid_all_max (every scene, no filter)": mx("bwd_kkt_resid_all_max"),
                "bwd_kkt_resid_all_scenes_over_1e-6": tot("bwd_kkt_resid_all_scenes_over_1e-6"),
                "bwd_input_sensitive_scenes (oracle's own dl/dp moves by more than 1e-4 under fp32 rounding of its inputs: not compared)": tot("bwd_input_sensitive_scenes"),
                "bwd_input_sensitivity_max": mx("bwd_input_sensitivity_max"), "fwd_input_sensitivity_err_x_max": mx("fwd_input_sensitivity_err_x_max"), "wall_s": time.time() - t}
print(json.dumps(out))

"""Turn a tools/pmc_summary.py table into the two small JSON files bench.py quotes in its `roofline` object:
    python tools/make_profile_json.py profiles/r03_pmc.txt r03
writes profiles/r02_traffic.json (FETCH_SIZE / WRITE_SIZE, KB per launch of the forward kernel) and profiles/r02_counters.json
(VALU-active and wait fractions of the wave cycles), keyed like bench.py keys its configurations."""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# bench.py's configuration key -> (a substring of) the mangled kernel name in the rocprofv3 tables
KERNELS = {"fused_B4096_nc16_f64": "lcp_fwd_quadIfdLb1ELi1ELi2E",                 # the headline forward (body space, pinned floor)
           "fused_B4096_nc16_f64_bwd": "lcp_bwd_quadIfdLb1E",                      # the dense backward behind it (body space)
           "fused_B4096_nc16_f64_bwd_physical": "lcp_bwd_step_quadIfdLi1ELb1E",    # --bwd physical
           "dense_B4096_nc16_f64": "lcp_fwd_quadIfdLb0ELi1E",
           "dense_B4096_nc16_f64_bwd": "lcp_bwd_quadIfdLb0E"}


def main(path, tag):
    rows = {}
    for line in open(path):
        parts = line.split()
        if len(parts) >= 5 and parts[0].startswith("_ZN"):
            rows[(parts[0], parts[1])] = float(parts[-1])          # avg per launch (per counter instance)
    traffic, counters = {}, {}
    src = os.path.relpath(path, ROOT)
    for key, sub in KERNELS.items():
        get = lambda c: next((v for (k, cn), v in rows.items() if sub in k and cn == c), None)
        if get("FETCH_SIZE") is not None and get("WRITE_SIZE") is not None:
            traffic[key] = {"kernel": sub, "fetch_kb": get("FETCH_SIZE"), "write_kb": get("WRITE_SIZE"),
                            "source": src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, avg per launch)"}
        wc = get("SQ_WAVE_CYCLES")
        if wc:
            c = {"kernel": sub, "source": src + " (rocprofv3 --pmc, avg per launch and counter instance)"}
            if get("SQ_ACTIVE_INST_VALU") is not None:
                c["valu_active"] = get("SQ_ACTIVE_INST_VALU") / wc
            if get("SQ_WAIT_ANY") is not None:
                c["wait_frac"] = get("SQ_WAIT_ANY") / wc
            if get("SQ_INSTS_VALU") is not None and get("SQ_WAVES"):
                c["valu_insts_per_wave"] = get("SQ_INSTS_VALU") / get("SQ_WAVES")
            c["mfma_f64_ops"] = get("SQ_INSTS_VALU_MFMA_MOPS_F64")
            counters[key] = c
    json.dump(traffic, open(os.path.join(ROOT, "profiles", tag + "_traffic.json"), "w"), indent=1)
    json.dump(counters, open(os.path.join(ROOT, "profiles", tag + "_counters.json"), "w"), indent=1)
    print(json.dumps({"traffic": traffic, "counters": counters}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

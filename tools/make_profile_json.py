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
           # --mode dense (round 4: body space): the forward CALL is classify + the pinned kernel + three launches that find no scene
           # (general body-space kernel, the wave-per-scene fallbacks - one launch since the merge of round 4) - traffic summed over all of them, counters of the solver
           "dense_B4096_nc16_f64": ["lcp_fwd_quadIfdLb0ELi1ELi2E", "lcp_classify_wave", "lcp_fwd_quadIfdLb0ELi1ELi1E", "lcp_fwd_wave_any"],
           "dense_B4096_nc16_f64_bwd": ["lcp_bwd_quadIfdLb1E", "lcp_bwd_wave_any"],
           # --config 4 (the piles): lcp_primal_kernel<30, fwd, PIN = 3> and its backward <32, bwd, PIN = 3>
           "fused_B4096_nc64_f64": "lcp_primal_kernelILi30ELb0ELb0ELi4ELi3E",
           "fused_B4096_nc64_f64_bwd_physical": "lcp_primal_kernelILi32ELb1ELb0ELi4ELi3E",
           # --config 4 --mode dense: classify (+ extract) + lcp_primal_kernel<30, fwd, DENSE, PIN> (the scenes whose rows pin the floor: class 4) +
           # the general body-space, contact-space and generic launches that find no scene; --contact-space: lcp_big_kernel<64, fwd, DENSE> does the work.
           # ("@part": only tables whose FILE name contains `part` - the two runs launch the same kernel names)
           "dense_B4096_nc64_f64": ["lcp_primal_kernelILi30ELb0ELb1ELi4ELi3E", "lcp_classify_big", "lcp_primal_kernelILi40ELb0ELb1E", "lcp_big_kernelILi64ELb0ELb1E", "lcp_fwd_kernel", "@dense5."],
           "dense_B4096_nc64_f64_bwd": ["lcp_primal_kernelILi32ELb1ELb1ELi4ELi3E", "lcp_primal_kernelILi40ELb1ELb1E", "lcp_big_kernelILi64ELb1ELb1E", "lcp_bwd_kernel", "@dense5."],
           "dense_cs_B4096_nc64_f64": ["lcp_big_kernelILi64ELb0ELb1E", "lcp_classify_big", "lcp_fwd_kernel", "@dense5cs."],
           "dense_cs_B4096_nc64_f64_bwd": ["lcp_big_kernelILi64ELb1ELb1E", "lcp_bwd_kernel", "@dense5cs."]}


def main(paths, tag):
    rows = {}
    path = paths[0]
    for p_ in paths:
        for line in open(p_):
            parts = line.split()
            if len(parts) >= 5 and parts[0].startswith("_ZN"):
                rows[(os.path.basename(p_), parts[0], parts[1])] = float(parts[-1])          # avg per launch (per counter instance)
    traffic, counters = {}, {}
    for key, subs in KERNELS.items():
        subs = [subs] if isinstance(subs, str) else list(subs)
        part = next((sb[1:] for sb in subs if sb.startswith("@")), None)   # restrict to the tables of one run
        subs = [sb for sb in subs if not sb.startswith("@")]
        sub = subs[0]                                                   # the kernel the counters describe; traffic sums the whole call
        files = sorted({f for (f, _, _) in rows if (part is None or part in f)})
        if part is None:
            files = [f for f in files if "dense5" not in f]
        src = ", ".join("profiles/" + f for f in files)
        get = lambda c, sb=sub: next((v for (f, k, cn), v in rows.items() if f in files and sb in k and cn == c), None)
        if get("FETCH_SIZE") is not None and get("WRITE_SIZE") is not None:
            tot = lambda c: sum(v for v in (get(c, sb) for sb in subs) if v is not None)
            traffic[key] = {"kernel": sub, "fetch_kb": tot("FETCH_SIZE"), "write_kb": tot("WRITE_SIZE"),
                            "source": src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, avg per launch"
                                      + ("; summed over the %d kernels of the call" % len(subs) if len(subs) > 1 else "") + ")"}
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
            if get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and get("SQ_BUSY_CYCLES"):
                c["mfma_busy_frac"] = get("SQ_VALU_MFMA_BUSY_CYCLES") / get("SQ_BUSY_CYCLES")
            counters[key] = c
    sys.path.insert(0, ROOT)
    from lcp_physics_amd.srchash import source_sha256
    stamp = source_sha256()                              # (bench.py prints "counters_stale" when the kernel sources have moved on since)
    traffic["source_sha256"] = stamp
    counters["source_sha256"] = stamp
    json.dump(traffic, open(os.path.join(ROOT, "profiles", tag + "_traffic.json"), "w"), indent=1)
    json.dump(counters, open(os.path.join(ROOT, "profiles", tag + "_counters.json"), "w"), indent=1)
    print(json.dumps({"traffic": traffic, "counters": counters}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:-1], sys.argv[-1])          # python tools/make_profile_json.py profiles/r04_pmc.txt [profiles/r04_pmc_dense.txt ...] r04

"""Register / scratch / occupancy report of every kernel in lcp_physics_amd/csrc, from the compiler itself
(`hipcc -Rpass-analysis=kernel-resource-usage`; the vgpr / agpr columns of a rocprofv3 kernel trace are not the allocation).
Runs without a GPU.    python tools/kernel_resources.py > profiles/r02_kernel_resources.json"""
import json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lcp_physics_amd", "csrc")
FILES = {"lcp_quad.hip": ["-fno-slp-vectorize"], "lcp_quad_n15e3.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-memory-clause"], "lcp_quad_n9e3.hip": ["-fno-slp-vectorize"], "lcp_quad_n12e3.hip": ["-fno-slp-vectorize"], "lcp_quad_n6e3.hip": ["-fno-slp-vectorize"],
         "lcp_solo.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"], "lcp_big.hip": [], "lcp_primal.hip": [], "lcp_primal_pin.hip": [], "lcp_primal_chain.hip": [], "lcp_primal_poststab.hip": [], "lcp_wave64.hip": [], "lcp_generic.hip": [], "lcp_contacts.hip": []}
KEYS = {"VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
        "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "LDS Size [bytes/block]": "lds_static_bytes",
        "VGPRs Spill": "vgpr_spills", "SGPRs Spill": "sgpr_spills", "TotalSGPRs": "sgpr"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    table = {}
    for f, extra in FILES.items():
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
               "-x", "hip", "-c", os.path.join(CSRC, f), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
        err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
            if m:
                cur = m.group(1); table[cur] = {"file": f}
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
            if m and cur and m.group(1).strip() in KEYS:
                table[cur][KEYS[m.group(1).strip()]] = int(m.group(2))
    names = demangle(list(table))
    nice = {names[k]: v for k, v in table.items()}
    pick = lambda sub: next((dict(v, kernel=k) for k, v in nice.items() if sub in k), None)
    out = {
        # the keys bench.py quotes in its roofline object
        "lcp_fwd_quad_f64_fused": pick("lcp_fwd_quad<float, double, true, 1, 2, 15, 3, 16, true, false>"),   # the headline (<= 4096 scenes): sizes at compile time, Jc / Jt columns in registers
        "lcp_fwd_quad_f64_fused_two_waves": pick("lcp_fwd_quad<float, double, true, 1, 2, 15, 3, 16, false, false>"),   # > 4096 scenes: the lean form, two waves per SIMD
        "lcp_fwd_quad_f64_fused_runtime_sizes": pick("lcp_fwd_quad<float, double, true, 1, 2, 0, 0, 0, false, false>"),
        "lcp_fwd_quad_f64_post_stab": pick("lcp_fwd_quad<float, double, true, 1, 2, 0, 0, 0, false, true>"),   # post-stabilisation on the four-scenes mapping (round 5)
        "lcp_bwd_quad_f64_body": pick("lcp_bwd_quad<float, double, true, 15, 3, true>"),
        "lcp_bwd_step_quad_f64_body": pick("lcp_bwd_step_quad<float, double, 1, true, 15, 3, true>"),
        "lcp_fwd_solo_9_3_8": pick("lcp_fwd_solo<9, 3, 8>"),
        "lcp_fwd_quad_f64_fused_general_equality_rows": pick("lcp_fwd_quad<float, double, true, 1, 1, 0, 0, 0, false, false>"),
        "lcp_fwd_quad_f64_fused_contact_space": pick("lcp_fwd_quad<float, double, true, 1, 0, 0, 0, 0, false, false>"),
        "lcp_fwd_quad_f64_dense": pick("lcp_fwd_quad<float, double, false, 1, 2, 15, 3, 16, true, false>"),   # the dense boundary, body space (round 4): the same kernel reading rows of G
        "lcp_fwd_quad_f64_dense_contact_space": pick("lcp_fwd_quad<float, double, false, 1, 0, 0, 0, 0, false, false>"),   # LCP_PATH_CONTACT_SPACE
        "lcp_fwd_quad_f32_fused": pick("lcp_fwd_quad<float, float, true, 1, 0, 0, 0, 0, false, false>"),
        "lcp_fwd_quad_f32_dense": pick("lcp_fwd_quad<float, float, false, 1, 0, 0, 0, 0, false, false>"),
        "lcp_big_kernel_64_fwd": pick("lcp_big_kernel<64, false, false>"),
        "lcp_primal_kernel_30_pinned_fwd": pick("lcp_primal_kernel<30, false, false, 4, 3>"),   # BASELINE config 5 under LCP_HINT_PINNED
        "lcp_primal_kernel_32_pinned_bwd": pick("lcp_primal_kernel<32, true, false, 4, 3>"),
        "lcp_primal_kernel_dense_fwd": pick("lcp_primal_kernel<30, false, true, 4, 3>"),        # configs[4] through the dense boundary: the pinned form reading lcp_classify_big's records
        "lcp_primal_kernel_dense_bwd": pick("lcp_primal_kernel<32, true, true, 4, 3>"),
        "lcp_big_kernel_64_dense_fwd": pick("lcp_big_kernel<64, false, true>"),                 # ... with LCP_PATH_CONTACT_SPACE
        "lcp_classify_big": pick("lcp_classify_big"),
        "lcp_primal_kernel_40_fwd": pick("lcp_primal_kernel<40, false, false, 4, 0>"),           # the general form (36 x 36 on config 5)
        "lcp_primal_kernel_40_bwd": pick("lcp_primal_kernel<40, true, false, 4, 0>"),
        "all_kernels": nice,
        "source": "hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py)",
    }
    sys.path.insert(0, ROOT)
    from lcp_physics_amd.srchash import source_sha256
    out["source_sha256"] = source_sha256()               # (bench.py prints "counters_stale" when the kernel sources have moved on since)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

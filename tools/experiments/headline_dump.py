"""Two halves of one measurement, so that the GPU box only does GPU work:

  dump   (on the GPU box)   python tools/experiments/headline_dump.py dump  gpurun_out/r05_dump [label ...]
         runs the kernel pairs of tests/test_hip_headline_parity.py's cases (run_kernels: no oracle) and stores what they wrote -
         x, z, s, iterations, status, dp, dh, db, dQ, dA, the physical gradients - as .npz.  dG and dF are rank-1 / rank-2 in those
         (lcp.py:53-54: dG = dlam x^T + lam dx^T, dF = -dlam lam^T with dlam = -dh): checked on the device against what the kernel
         stored (`rank1_err_dG`, `rank1_err_dF`, relative to the largest entry) instead of being carried home (dF alone is 256 KB a pile).
  report (in the build container, CPU)   python tools/experiments/headline_dump.py report gpurun_out/r05_dump [label ...]
         rebuilds the same scenes from their seeds, assembles the LCPs with the oracle's restatement of engines.py:50-74 in fp32 (entry for
         entry what the HIP assembly produces: bench.py's parity.assembly_max_rel_diff_vs_oracle_assembly = 0), rebuilds dG / dF from
         the rank-1 factors, and prints tests/parity.py::headline_report for each case - the fp64 oracle runs here, not on the GPU box.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# label -> (run_kernels kwargs, scenes kept)
CASES = {
    "configs1_1024x8": (dict(kind="stack", B=1024, nbox=2, seed=1236), 1024),
    "configs2_4096x16": (dict(kind="stack", B=4096, nbox=4, seed=1236), 4096),
    "configs2_4096x16_count": (dict(kind="stack", B=4096, nbox=4, seed=1236, entry="count"), 4096),
    "configs1_1024x8_count": (dict(kind="stack", B=1024, nbox=2, seed=1236, entry="count"), 1024),
    "configs2_4096x16_dense": (dict(kind="stack", B=4096, nbox=4, seed=1236, entry="dense"), 4096),
    "configs2_4096x8_two_points": (dict(kind="stack", B=4096, nbox=4, seed=1236, pts=2, both_backwards=True), 4096),
    "configs1_1024x4_two_points": (dict(kind="stack", B=1024, nbox=2, seed=1236, pts=2, both_backwards=True), 1024),
    "configs2_4096x4_one_point": (dict(kind="stack", B=4096, nbox=4, seed=1236, pts=1, both_backwards=True), 4096),
    "configs4_4096x64_pile": (dict(kind="pile", B=4096, nbox=10, seed=5), 4096),
    "configs4_4096x64_pile_dense": (dict(kind="pile", B=4096, nbox=10, seed=5, entry="dense"), 1024),
    "configs4_4096x64_pile_dense_contact_space": (dict(kind="pile", B=4096, nbox=10, seed=5, entry="dense", path="big"), 1024),
}


def dump(outdir, labels):
    from tests.test_hip_headline_parity import run_kernels
    os.makedirs(outdir, exist_ok=True)
    for label in labels:
        kw, keep = CASES[label]
        K = run_kernels(**kw)
        B = K["sc"].B
        idx = torch.arange(0, B, max(1, B // keep))[:keep].to("cuda")
        rec = {"idx": idx.cpu().numpy(), "cot": K["cot"].numpy()}
        for k in ("x", "z", "s", "iters", "status"):
            rec[k] = K[k][idx].cpu().numpy()
        if K["grads"] is not None:
            g = dict(zip("QpGhAbF", K["grads"]))
            for k in ("phAb" if K["pile"] else "QphAb"):
                if g[k] is not None:
                    rec["d" + k] = g[k][idx].cpu().numpy()
            # the rank-1 structure of what stays behind (lcp.py:53-54, :59-60), whole batch
            x, z = K["x"].double(), K["z"].double()
            dx, dlam = g["p"].double(), -g["h"].double()
            dQ = 0.5 * (dx.unsqueeze(2) * x.unsqueeze(1) + x.unsqueeze(2) * dx.unsqueeze(1))
            rec["rank1_err_dQ"] = float(((g["Q"].double() - dQ).abs().amax(dim=(1, 2)) / dQ.abs().amax(dim=(1, 2)).clamp_min(1e-300)).max())
            dG = dlam.unsqueeze(2) * x.unsqueeze(1) + z.unsqueeze(2) * dx.unsqueeze(1)
            rec["rank1_err_dG"] = float(((g["G"].double() - dG).abs().amax(dim=(1, 2)) / dG.abs().amax(dim=(1, 2)).clamp_min(1e-300)).max())
            del dG
            worst = 0.0
            for lo in range(0, B, 256):                        # (dF in fp64: 512 KB per pile)
                sl = slice(lo, min(B, lo + 256))
                dF = -dlam[sl].unsqueeze(2) * z[sl].unsqueeze(1)
                worst = max(worst, float(((g["F"][sl].double() - dF).abs().amax(dim=(1, 2)) / dF.abs().amax(dim=(1, 2)).clamp_min(1e-300)).max()))
            rec["rank1_err_dF"] = worst
        if K["phys_grads"] is not None:
            for k, v in K["phys_grads"].items():
                rec["phys_" + k] = v[idx].cpu().numpy()
        np.savez_compressed(os.path.join(outdir, label + ".npz"), **rec)
        print(label, {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in rec.items()}, flush=True)
        del K
        torch.cuda.empty_cache()


def report(outdir, labels, **opts):
    from lcp_physics_amd import scenes
    from oracle import pdipm_oracle as O
    from tests import parity
    cache = {}
    for label in labels:
        path = os.path.join(outdir, label + ".npz")
        if not os.path.exists(path):
            print(label, "no dump")
            continue
        kw, keep = CASES[label]
        d = np.load(path)
        pile = kw["kind"] == "pile"
        sc = (scenes.make_pile_scenes(B=kw["B"], seed=kw["seed"], dtype=torch.float32) if pile else
              scenes.make_stack_scenes(B=kw["B"], nbox=kw["nbox"], pts_per_interface=kw.get("pts", 4), seed=kw["seed"], dtype=torch.float32))
        idx = torch.from_numpy(d["idx"])
        sub = sc.slice(0, sc.B)
        for fl in ("p", "v", "Mdiag", "f", "rest", "fric", "c_n", "c_p1", "c_p2", "c_i1", "c_i2", "Je"):
            setattr(sub, fl, getattr(sc, fl)[idx])
        lcp64 = [None if t is None else t.double() for t in O.assemble_lcp(*sub.assembly_args())]
        T = lambda k: torch.from_numpy(d[k])
        x, z, s = T("x"), T("z"), T("s")
        cot = T("cot")[idx]
        grads = None
        if "dp" in d:
            dx, dlam = T("dp").double(), -T("dh").double()
            dQ = T("dQ") if "dQ" in d else 0.5 * (dx.unsqueeze(2) * x.double().unsqueeze(1) + x.double().unsqueeze(2) * dx.unsqueeze(1))
            grads = {"Q": dQ, "p": T("dp"), "h": T("dh"), "A": T("dA") if "dA" in d else None, "b": T("db") if "db" in d else None,
                     "G": dlam.unsqueeze(2) * x.double().unsqueeze(1) + z.double().unsqueeze(2) * dx.unsqueeze(1),
                     "F": -dlam.unsqueeze(2) * z.double().unsqueeze(1)}
        pg = {k[5:]: T(k) for k in d.files if k.startswith("phys_")} or None
        ck = cache.setdefault((kw["kind"], kw["B"], kw["nbox"], kw["seed"], kw.get("pts", 4), len(idx)), {})
        rep, ref = parity.headline_report(O, lcp64, x, z, s, T("iters"), cot=cot, grads=grads, phys_grads=pg, phys=sub.phys_dict(), dt=sub.dt,
                                          cache=ck, **opts)
        rep["status_nonzero"] = int((T("status") & ~4 != 0).sum())
        for k in ("rank1_err_dG", "rank1_err_dF", "rank1_err_dQ"):
            if k in d:
                rep[k] = float(d[k])
        print("headline parity %s: %s" % (label, json.dumps(rep)), flush=True)


if __name__ == "__main__":
    mode, outdir = sys.argv[1], sys.argv[2]
    labels = sys.argv[3:] or list(CASES)
    if mode == "dump":
        dump(outdir, labels)
    else:
        report(outdir, labels, input_stability="--input-stability" in os.environ.get("HEADLINE_OPTS", ""),
               all_grads="--all-grads" in os.environ.get("HEADLINE_OPTS", ""))

"""fp64-I/O operator boundary, converged solves: the kernel's backward against the ORACLE'S backward evaluated at the KERNEL'S own iterate
(same linear system, two solvers), per kernel path.  Diagnostic behind tests/test_hip_parity.py::test_fp64_io_takes_the_fast_kernels...
    python tools/experiments/fp64_io_backward_diag.py [path ...]      (paths: auto generic big)"""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lcp_physics_amd import _lib, scenes
from lcp_physics_amd.lcp import lcp_backward, lcp_solve
from oracle import pdipm_oracle as O
from tests import parity

dev = torch.device("cuda")
sc = scenes.make_stack_scenes(B=64, nbox=4, pts_per_interface=4, seed=4242, dtype=torch.float64)
lcp64 = [None if t is None else t.double() for t in O.assemble_lcp(*sc.assembly_args())]
ref = O.lcp_forward(*lcp64)
cot = torch.randn(64, lcp64[0].shape[1], generator=torch.Generator().manual_seed(3), dtype=torch.float64)
gref = O.lcp_backward(ref, *lcp64, cot)
fl = parity.grad_floors(lcp64[0], lcp64[1], cot, ref.x, ref.z, ref.y)
for path in (sys.argv[1:] or ["auto", "generic"]):
    _lib.set_path(path)
    sol = lcp_solve(*[None if t is None else t.to(dev) for t in lcp64], compute="f64")
    grads = lcp_backward(sol, cot.to(dev))
    torch.cuda.synchronize()
    g64 = {k: (None if g is None else g.double().cpu()) for k, g in zip("QpGhAbF", grads)}
    at = copy.copy(ref)
    at.x, at.z, at.s, at.y = sol.x.double().cpu(), sol.z.double().cpu(), sol.s.double().cpu(), sol.y.double().cpu()
    gat = O.lcp_backward(at, *lcp64, cot)
    ok_at = parity.backward_well_posed(lcp64[0], lcp64[2], lcp64[4], lcp64[6], at, cot, gat)
    e_at = parity.err_grads({k: g64[k] for k in "QpAb"}, {k: gat["d" + k] for k in "QpAb"}, fl)
    e_ref = parity.err_grads({k: g64[k] for k in "QpAb"}, {k: gref["d" + k] for k in "QpAb"}, fl)
    same = (parity._n(at.z - ref.z) / parity._n(ref.z)) < 2e-3
    rk = parity.kkt_backward_residual(lcp64[0], lcp64[2], lcp64[4], lcp64[6], at.z, at.s, cot, g64["p"], -g64["h"], -g64["b"])
    ro = parity.kkt_backward_residual(lcp64[0], lcp64[2], lcp64[4], lcp64[6], at.z, at.s, cot, gat["dp"], -gat["dh"], -gat["db"])
    print("path", path, " iters kernel", sol.iters.cpu().tolist()[:16], " oracle", ref.iters.tolist()[:16])
    print("  well-posed at the kernel's iterate:", int(ok_at.sum()), " same iterate as the oracle's:", int(same.sum()))
    bad = torch.nonzero(e_at["p"] > 1e-6).flatten().tolist()
    print("  scenes with |dp - oracle_at_kernel_iterate| > 1e-6:", len(bad))
    for i in bad[:12]:
        print("   scene %2d same %d ok_at %d  err vs oracle@kernel-iterate %.2e  vs oracle@oracle-iterate %.2e | kernel resid dual %.1e ineq %.1e eq %.1e | oracle@kernel resid %.1e %.1e | min z %.1e min s %.1e  |z-zref|/|zref| %.1e"
              % (i, int(same[i]), int(ok_at[i]), float(e_at["p"][i]), float(e_ref["p"][i]), float(rk["dual"][i]), float(rk["ineq"][i]), float(rk["eq"][i]),
                 float(ro["dual"][i]), float(ro["ineq"][i]), float(at.z[i].min()), float(at.s[i].min()), float(parity._n(at.z - ref.z)[i] / parity._n(ref.z)[i])))
    print("  max err vs oracle@kernel-iterate, all scenes:", {k: float(v.max()) for k, v in e_at.items()})
    if path == "auto" and os.environ.get("LCP_DIAG_SCENE"):
        i = int(os.environ["LCP_DIAG_SCENE"])
        torch.set_printoptions(precision=3, linewidth=220, sci_mode=True)
        nc = at.z.shape[1] // 4
        print("scene", i, "nc", nc)
        for name, t in (("z", at.z[i]), ("s", at.s[i]), ("s/z", at.s[i] / at.z[i]), ("dlam kernel", -g64["h"][i]), ("dlam oracle@kernel", -gat["dh"][i]),
                        ("dx kernel", g64["p"][i]), ("dx oracle@kernel", gat["dp"][i]), ("dnu kernel", -g64["b"][i]), ("dnu oracle@kernel", -gat["db"][i])):
            print(" ", name, t.reshape(-1, nc) if t.numel() == 4 * nc else t)
        # row-wise residual of the inequality block for both
        G, F = lcp64[2][i], lcp64[6][i]
        for name, dx, dl in (("kernel", g64["p"][i], -g64["h"][i]), ("oracle", gat["dp"][i], -gat["dh"][i])):
            r3 = G @ dx - (at.s[i] / at.z[i]) * dl - F @ dl
            print("  ineq residual rows", name, r3.reshape(-1, nc))

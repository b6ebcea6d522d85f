"""Diagnostic (GPU): the fp64-I/O dense solve of tests/test_hip_parity.py::test_fp64_io_takes_the_fast_kernels... per scene - distance of the returned
iterate from the oracle's, the oracle's residual history (which iterate it keeps as best and by what margin), the backward errors.
    [LCP_HIP_LIB=...] python tools/experiments/best_iterate_ties.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pdipm_oracle as O
from tests import parity
from lcp_physics_amd import scenes
from lcp_physics_amd.lcp import lcp_backward, lcp_solve
sc = scenes.make_stack_scenes(B=64, nbox=4, pts_per_interface=4, seed=4242, dtype=torch.float64)
lcp64 = [None if t is None else t.double() for t in O.assemble_lcp(*sc.assembly_args())]
trace = []
ref = O.lcp_forward(*lcp64, trace=trace)
sol = lcp_solve(*[None if t is None else t.cuda().contiguous() for t in lcp64])
cot = torch.randn(64, lcp64[0].shape[1], generator=torch.Generator().manual_seed(3), dtype=torch.float64)
grads = lcp_backward(sol, cot.cuda())
gref = O.lcp_backward(ref, *lcp64, cot)
ok = parity.backward_well_posed(lcp64[0], lcp64[2], lcp64[4], lcp64[6], ref, cot, gref)
fl = parity.grad_floors(lcp64[0], lcp64[1], cot, ref.x, ref.z, ref.y)
errs = parity.err_grads({"p": grads[1].cpu()}, {"p": gref["dp"]}, fl)["p"]
ex = parity.err_x(sol.x.cpu(), ref.x, lcp64[0], lcp64[1])
relz = parity._n(sol.z.cpu() - ref.z) / parity._n(ref.z)
bad = torch.argsort(errs * ok, descending=True)[:4]
for k in bad.tolist():
    hist = [float(t["resid"][k]) for t in trace]
    print("scene %2d ok %s  dp err %.2e  err_x %.2e  rel|z - z_ref| %.2e  iters %d / %d  oracle residuals per iteration %s" % (
        k, bool(ok[k]), float(errs[k]), float(ex[k]), float(relz[k]), int(sol.iters[k]), int(ref.iters[k]), None if hist is None else ["%.3e" % h for h in hist]))
torch.save({"z": sol.z.cpu(), "x": sol.x.cpu()}, "/tmp/ties_%s.pt" % ("exp" if os.environ.get("LCP_HIP_LIB") else "head"))
if os.path.exists("/tmp/ties_head.pt") and os.path.exists("/tmp/ties_exp.pt"):
    a, b = torch.load("/tmp/ties_head.pt"), torch.load("/tmp/ties_exp.pt")
    d = parity._n(a["z"] - b["z"]) / parity._n(a["z"])
    print("HEAD against the other build, rel|z - z'| per scene: max %.2e, scenes above 1e-6: %s" % (float(d.max()), (d > 1e-6).nonzero().flatten().tolist()))

"""Dense LCPFunction boundary on CONVERGED solves, any kernel path: the kernel's backward against the oracle's backward at the kernel's
own iterate (tests/parity.py::own_iterate_backward) - where does a backward kernel divide by rounding noise ?
    python tools/experiments/own_iterate_diag.py NBOX PTS B DTYPE path [path ...]      e.g.  5 4 256 float32 auto big generic"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lcp_physics_amd import _lib, scenes
from lcp_physics_amd.lcp import lcp_backward, lcp_solve
from oracle import pdipm_oracle as O
from tests import parity

nbox, pts, B, dtype = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), getattr(torch, sys.argv[4])
dev = torch.device("cuda")
sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=4242, dtype=dtype)
lcp = O.assemble_lcp(*sc.assembly_args())
if os.environ.get("LCP_DIAG_DENSEQ"):        # a non-diagonal SPD Q: contact structure without the diagonal Q the quad kernels want -> lcp_*_wave_any
    Q = lcp[0]
    gq = torch.Generator().manual_seed(5)
    E = torch.randn(Q.shape, generator=gq, dtype=Q.dtype) * 0.02
    d = torch.diagonal(Q, dim1=1, dim2=2).sqrt()
    lcp = list(lcp)
    lcp[0] = Q + (E + E.transpose(1, 2)) * d.unsqueeze(2) * d.unsqueeze(1)
lcp64 = [None if t is None else t.double() for t in lcp]
ref = O.lcp_forward(*lcp64)
cot = torch.randn(B, lcp64[0].shape[1], generator=torch.Generator().manual_seed(3), dtype=torch.float64)
fl = parity.grad_floors(lcp64[0], lcp64[1], cot, ref.x, ref.z, ref.y)
print("scenes: %d boxes x %d points, nz %d nineq %d, %s; oracle iterations %s" % (nbox, pts, lcp64[0].shape[1], lcp64[2].shape[1], sys.argv[4],
      torch.bincount(ref.iters.to(torch.int64)).tolist()))
for path in sys.argv[5:]:
    _lib.set_path(path)
    sol = lcp_solve(*[None if t is None else t.to(dev) for t in lcp], compute="f64")
    grads = lcp_backward(sol, cot.to(dev=dev, dtype=dtype) if False else cot.to(dev).to(dtype))
    torch.cuda.synchronize()
    g64 = {k: (None if g is None else g.double().cpu()) for k, g in zip("QpGhAbF", grads)}
    rep = parity.own_iterate_backward(O, lcp64, ref, cot, sol.x.double().cpu(), sol.z.double().cpu(), sol.s.double().cpu(), g64, fl)
    fin = all(bool(torch.isfinite(g).all()) for g in g64.values() if g is not None)
    print("path %-8s finite %d  |dh| max %.2e  %s" % (path, fin, float(g64["h"].abs().max()), rep))

import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lcp_physics_amd import _lib, scenes
from lcp_physics_amd.lcp import lcp_backward, lcp_solve
from lcp_physics_amd.physics import fused_step
from oracle import pdipm_oracle as O
from tests import parity
torch.set_printoptions(precision=5, linewidth=220)
DEV = "cuda"
nbox, pts, B = 2, 4, 64
sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=99 + nbox, dtype=torch.float32)
lcp32 = O.assemble_lcp(*sc.assembly_args())
lcp64 = [None if t is None else t.double() for t in lcp32]
ref = O.lcp_forward(*lcp64)
g = torch.Generator().manual_seed(5)
cot = torch.randn(B, lcp32[0].shape[1], generator=g, dtype=torch.float32)
gref = O.lcp_backward(ref, *lcp64, cot.double())
gref = {k: gref["d" + k] for k in "QpGhAbF"}
for path in ("wave64", "generic"):
    _lib.set_path(path)
    sol = lcp_solve(*[None if t is None else t.to(DEV) for t in lcp32])
    grads = lcp_backward(sol, cot.to(DEV))
    torch.cuda.synchronize()
    grads = {k: (None if t is None else t.double().cpu()) for k, t in zip("QpGhAbF", grads)}
    Q, p = lcp64[0], lcp64[1]
    fl = parity.grad_floors(Q, p, cot.double(), ref.x, ref.z, ref.y)
    errs = parity.err_grads(grads, gref, fl)
    ph = {k: v.double() if v.is_floating_point() else v for k, v in sc.phys_dict().items()}
    pg = parity.physical_grads(ph, sc.dt, grads, O)
    pg_ref = parity.physical_grads(ph, sc.dt, gref, O)
    scl = parity.free_scales(Q, p, cot.double())
    floor = parity._n(cot) * torch.maximum(scl["x_free"], parity._n(ref.x))
    ep = parity.err_physical(pg, pg_ref, ph, floor)
    i = int(ep.argmax())
    print("==", path, "phys err max %.3e at scene %d" % (float(ep.max()), i), "n>1e-4:", int((ep > 1e-4).sum()))
    print("   direct errs at scene:", {k: "%.2e" % float(v[i]) for k, v in errs.items()})
    print("   iters hip/oracle", int(sol.iters[i]), int(ref.iters[i]), "status", int(sol.status[i]), "ref resid %.2e" % float(ref.resid[i]))
    print("   z hip ", sol.z[i].double().cpu()); print("   z ref ", ref.z[i])
    print("   s hip ", sol.s[i].double().cpu()); print("   s ref ", ref.s[i])
    print("   dp hip", grads["p"][i]); print("   dp ref", gref["p"][i])
    print("   dh hip", grads["h"][i]); print("   dh ref", gref["h"][i])
    for k in parity.PHYS_KEYS:
        print("   phys", k, "diff %.3e ref %.3e" % (float((pg[k][i] - pg_ref[k][i]).norm()), float(pg_ref[k][i].norm())))
    res = parity.kkt_backward_residual(lcp64[0], lcp64[2], lcp64[4], lcp64[6], sol.z.double().cpu(), sol.s.double().cpu(), cot.double(), grads["p"], -grads["h"], -grads["b"])
    print("   kkt residual at scene", {k: "%.2e" % float(v[i]) for k, v in res.items()})
    res = parity.kkt_backward_residual(lcp64[0], lcp64[2], lcp64[4], lcp64[6], ref.z, ref.s, cot.double(), gref["p"], -gref["h"], -gref["b"])
    print("   oracle kkt residual   ", {k: "%.2e" % float(v[i]) for k, v in res.items()})
# fused active-set failure
_lib.set_path("wave64")
sc = scenes.make_stack_scenes(B=128, nbox=2, pts_per_interface=4, seed=21, dtype=torch.float32)
sc64 = sc.to(dtype=torch.float64)
new_v, ref, lcp = O.solve_dynamics(*sc64.assembly_args())
out = fused_step(sc.to(device=DEV)); torch.cuda.synchronize()
z, s = out["z"].double().cpu(), out["s"].double().cpu()
bad = torch.nonzero(parity.active_sets(z, s) != parity.active_sets(ref.z, ref.s))
print("fused mismatches", bad.shape[0])
for b_, i_ in bad.tolist()[:12]:
    print("   scene %d idx %d  hip z %.3e s %.3e | ref z %.3e s %.3e | zmax %.2e smax %.2e" % (b_, i_, z[b_, i_], s[b_, i_], ref.z[b_, i_], ref.s[b_, i_], ref.z[b_].max(), ref.s[b_].max()))

import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lcp_physics_amd import _lib, scenes
from lcp_physics_amd.lcp import lcp_backward, lcp_solve
from oracle import pdipm_oracle as O
from tests import parity
torch.set_printoptions(precision=5, linewidth=220)
DEV = "cuda"
nbox, pts, B = 4, 2, 64
sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=99 + nbox, dtype=torch.float32)
lcp32 = O.assemble_lcp(*sc.assembly_args())
lcp64 = [None if t is None else t.double() for t in lcp32]
ref = O.lcp_forward(*lcp64)
g = torch.Generator().manual_seed(5)
cot = torch.randn(B, lcp32[0].shape[1], generator=g, dtype=torch.float32)
gref = O.lcp_backward(ref, *lcp64, cot.double())
gref = {k: gref["d" + k] for k in "QpGhAbF"}
Q, p, G, h, A, b, F = lcp64
res_o = parity.kkt_backward_residual(Q, G, A, F, ref.z, ref.s, cot.double(), gref["p"], -gref["h"], -gref["b"])
ok = torch.stack([v for v in res_o.values()]).max(dim=0)[0] < 1e-9
zs = ref.z.max(dim=1, keepdim=True)[0]; ss = ref.s.max(dim=1, keepdim=True)[0]
sc_margin = torch.maximum(ref.z / zs, ref.s / ss).min(dim=1)[0]
for path in ("generic", "wave64"):
    _lib.set_path(path)
    sol = lcp_solve(*[None if t is None else t.to(DEV) for t in lcp32])
    grads = lcp_backward(sol, cot.to(DEV)); torch.cuda.synchronize()
    grads = {k: (None if t is None else t.double().cpu()) for k, t in zip("QpGhAbF", grads)}
    ph = {k: v.double() if v.is_floating_point() else v for k, v in sc.phys_dict().items()}
    pg = parity.physical_grads(ph, sc.dt, grads, O); pg_ref = parity.physical_grads(ph, sc.dt, gref, O)
    scl = parity.free_scales(Q, p, cot.double())
    floor = parity._n(cot) * torch.maximum(scl["x_free"], parity._n(ref.x))
    ep = parity.err_physical(pg, pg_ref, ph, floor)
    epm = ep.clone(); epm[~ok] = 0
    i = int(epm.argmax())
    print("==", path, "worst ok scene", i, "ep %.3e" % float(ep[i]), "ok", int(ok.sum()), "margin %.2e" % float(sc_margin[i]), "iters", int(sol.iters[i]), int(ref.iters[i]), "status", int(sol.status[i]), "resid ref %.2e" % float(ref.resid[i]))
    for k in parity.PHYS_KEYS:
        print("   phys", k, "diff %.3e ref %.3e" % (float((pg[k][i] - pg_ref[k][i]).norm()), float(pg_ref[k][i].norm())))
    fl = parity.grad_floors(Q, p, cot.double(), ref.x, ref.z, ref.y)
    errs = parity.err_grads(grads, gref, fl)
    print("   direct", {k: "%.2e" % float(v[i]) for k, v in errs.items()})
    print("   z hip", sol.z[i].double().cpu()); print("   z ref", ref.z[i]); print("   s hip", sol.s[i].double().cpu()); print("   s ref", ref.s[i])
    print("   dh hip", grads["h"][i]); print("   dh ref", gref["h"][i])
    print("   ep per scene (ok only) top:", sorted([(float(ep[j]), j) for j in range(B) if ok[j]], reverse=True)[:6])

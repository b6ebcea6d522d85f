"""Diagnostic (GPU): the scenes behind the tails of the headline backward report - per scene: errors of dp / dQ / the physical
gradients, the iteration-count delta, how far the kernel's iterate is from the oracle's, the oracle's own complementarity margin and
backward residual.   python tools/experiments/bwd_outliers.py stack 4096 4 1236 pinned | stack 1024 2 1236 coupled [dense]"""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pdipm_oracle as O
from tests import parity
from lcp_physics_amd import scenes
from lcp_physics_amd.lcp import lcp_backward, lcp_solve
from lcp_physics_amd.physics import assemble_contacts, fused_step
from lcp_physics_amd.physics.batched_world import solution_of_step

kind, B, nbox, seed, rows = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
dense = len(sys.argv) > 6 and sys.argv[6] == "dense"
path = sys.argv[7] if len(sys.argv) > 7 else "auto"
DEV = "cuda"
sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=4, seed=seed, dtype=torch.float32)
if rows == "scaled": sc.Je = sc.Je * 2.0
elif rows == "coupled":
    sc.Je = sc.Je.clone(); sc.Je[:, 1, 3] = 0.25
scg = sc.to(device=DEV)
lcp = assemble_contacts(scg)
nz = 3 * sc.nb
cot = torch.randn(B, nz, generator=torch.Generator().manual_seed(4321), dtype=torch.float32)
if dense:
    sol = lcp_solve(*lcp, path=path)
    x, z, s, iters = sol.x, sol.z, sol.s, sol.iters
else:
    out = fused_step(scg)
    sol = solution_of_step(scg, out, lcp[2], lcp[4])
    x, z, s, iters = -out["v_new"].reshape(B, nz), out["z"], out["s"], out["iters"]
g7 = lcp_backward(sol, cot.to(DEV))
torch.cuda.synchronize()
lcp64 = [None if t is None else t.double().cpu() for t in lcp]
Q, p, G, h, A, b, F = lcp64
ref = O.lcp_forward(*lcp64)
c64 = cot.double()
gref = O.lcp_backward(ref, *lcp64, c64)
ok = parity.backward_well_posed(Q, G, A, F, ref, c64, gref)
fl = parity.grad_floors(Q, p, c64, ref.x, ref.z, ref.y)
g64 = {k: (None if t is None else t.double().cpu()) for k, t in zip("QpGhAbF", g7)}
gr = {k: gref["d" + k] for k in "QpGhAbF"}
errs = parity.err_grads({k: g64[k] for k in "QpAb"}, {k: gr[k] for k in "QpAb"}, fl)
ph = {k: (v.double() if (v is not None and v.is_floating_point()) else v) for k, v in sc.phys_dict().items()}
pg_ref = parity.physical_grads(ph, sc.dt, gr, O)
pg = parity.physical_grads(ph, sc.dt, {k: g64[k] for k in "QpGhF"}, O)
scl = parity.free_scales(Q, p, c64)
floor = parity._n(c64) * torch.maximum(scl["x_free"], parity._n(ref.x))
ep = {k: parity.err_physical(pg, pg_ref, ph, floor, keys=[k]) for k in ("Mdiag", "v", "f")}
epall = parity.err_physical(pg, pg_ref, ph, floor, keys=["Mdiag", "v", "f"])
d_it = iters.cpu().long() - ref.iters.long()
zk, sk = z.double().cpu(), s.double().cpu()
relz = parity._n(zk - ref.z) / parity._n(ref.z).clamp_min(1e-300)
zs, ss = ref.z.max(dim=1, keepdim=True)[0], ref.s.max(dim=1, keepdim=True)[0]
margin = torch.maximum(ref.z / zs, ref.s / ss).min(dim=1)[0]
res_o = parity.kkt_backward_residual(Q, G, A, F, ref.z, ref.s, c64, gref["dp"], -gref["dh"], None if gref.get("db") is None else -gref["db"])
res_o = torch.stack(list(res_o.values())).max(dim=0)[0]
res_k = parity.kkt_backward_residual(Q, G, A, F, zk, sk, c64, g64["p"], -g64["h"], None if g64.get("b") is None else -g64["b"])
res_k = torch.stack(list(res_k.values())).max(dim=0)[0]
mu_ref = (ref.z * ref.s).sum(dim=1) / ref.z.shape[1]
score = torch.where(ok, torch.maximum(errs["p"], epall), torch.zeros_like(epall))
top = torch.argsort(score, descending=True)[:8]
print("well-posed", int(ok.sum()), "of", B, " iters delta hist", {int(k): int((d_it == k).sum()) for k in torch.unique(d_it)})
print("well-posed AND iters differ:", int((ok & (d_it != 0)).sum()), " max dp err among those", float(errs["p"][ok & (d_it != 0)].max()) if bool((ok & (d_it != 0)).any()) else None,
      " max dp err among well-posed with equal iters", float(errs["p"][ok & (d_it == 0)].max()))
for k in top.tolist():
    print("scene %5d  dp %.2e dQ %.2e dA %.2e | phys Mdiag %.2e v %.2e f %.2e all %.2e | d_iters %+d  rel|z-z_ref| %.1e  margin %.1e  mu_ref %.1e  oracle kkt %.1e  kernel kkt(own) %.1e  |dlam_ref| %.1e |dlam_k| %.1e |dx_ref| %.1e" % (
        k, float(errs["p"][k]), float(errs["Q"][k]), float(errs["A"][k]) if "A" in errs else -1, float(ep["Mdiag"][k]), float(ep["v"][k]), float(ep["f"][k]), float(epall[k]),
        int(d_it[k]), float(relz[k]), float(margin[k]), float(mu_ref[k]), float(res_o[k]), float(res_k[k]), float(gref["dh"][k].norm()), float(g64["h"][k].norm()), float(gref["dp"][k].norm())))
# the same tails with the margin threshold of `backward_well_posed` raised: how the worst error depends on it
for thr in (1e-6, 1e-5, 1e-4, 1e-3):
    sel = ok & (margin > thr)
    print("margin > %.0e: %4d scenes  dp max %.2e  phys max %.2e" % (thr, int(sel.sum()), float(errs["p"][sel].max()), float(epall[sel].max())))
# how much the ORACLE moves on the same scenes when only its own arithmetic changes (LU without pivoting instead of with: pdipm.py
# factors with pivoting; the same equations): the scene's sensitivity, against which the kernel's distance is to be read
sub = [None if t is None else t[top] for t in lcp64]
ref_np = O.lcp_forward(*sub, pivot=False)
g_np = O.lcp_backward(ref_np, *sub, c64[top])
for i, k in enumerate(top.tolist()):
    ex_k = float(parity.err_x(x.double().cpu()[k:k + 1], ref.x[k:k + 1], Q[k:k + 1], p[k:k + 1]))
    ex_o = float(parity.err_x(ref_np.x[i:i + 1], ref.x[k:k + 1], Q[k:k + 1], p[k:k + 1]))
    dp_o = float(parity.err_grads({"p": g_np["dp"][i:i + 1]}, {"p": gref["dp"][k:k + 1]}, {kk: v[k:k + 1] for kk, v in fl.items()})["p"])
    print("scene %5d  err_x kernel vs oracle %.2e | oracle (no pivoting) vs oracle %.2e   dp: kernel %.2e | oracle variant %.2e   iters %d / %d / %d" % (
        k, ex_k, ex_o, float(errs["p"][k]), dp_o, int(iters[k]), int(ref.iters[k]), int(ref_np.iters[i])))
# ... and when the oracle's INPUTS change by fp32 rounding: the dense tensors above come from lcp_assemble_contacts_f32 (fp32 entries), the
# fused kernel assembles the same physical inputs in fp64 - the oracle on ITS OWN fp64 assembly of those inputs is the like-for-like reference
sct = sc.slice(0, B)
args64 = [a.double() if (isinstance(a, torch.Tensor) and a.is_floating_point()) else a for a in sct.assembly_args()]
sel = lambda a: a[top] if isinstance(a, torch.Tensor) else a
lcpo = [None if t is None else t for t in O.assemble_lcp(*[sel(a) for a in args64])]
ref_o = O.lcp_forward(*lcpo)
g_o = O.lcp_backward(ref_o, *lcpo, c64[top])
for i, k in enumerate(top.tolist()):
    Qo, po = lcpo[0][i:i + 1], lcpo[1][i:i + 1]
    ex_ko = float(parity.err_x(x.double().cpu()[k:k + 1], ref_o.x[i:i + 1], Qo, po))
    ex_oo = float(parity.err_x(ref.x[k:k + 1], ref_o.x[i:i + 1], Qo, po))
    flo = {kk: v[k:k + 1] for kk, v in fl.items()}
    dp_ko = float(parity.err_grads({"p": g64["p"][k:k + 1]}, {"p": g_o["dp"][i:i + 1]}, flo)["p"])
    dp_oo = float(parity.err_grads({"p": gref["dp"][k:k + 1]}, {"p": g_o["dp"][i:i + 1]}, flo)["p"])
    relz_ko = float(parity._n(zk[k:k + 1] - ref_o.z[i:i + 1]) / parity._n(ref_o.z[i:i + 1]))
    print("scene %5d  against the oracle on its own fp64 assembly: err_x kernel %.2e | oracle on the fp32 tensors %.2e   dp: kernel %.2e | oracle on the fp32 tensors %.2e   rel|z - z_ref| kernel %.1e" % (
        k, ex_ko, ex_oo, dp_ko, dp_oo, relz_ko))

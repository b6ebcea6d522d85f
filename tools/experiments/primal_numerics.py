"""Numerical experiment (CPU, fp64): PDIPM with the KKT solve done in BODY space,
    [[Q + G^T B G, A^T], [A, 0]] [dx; dy] = [-rx + G^T B q; -ry],  B = (F + diag(s/z))^-1 (4 x 4 blocks per contact),
    q = rs/d - rz,  dz = B (G dx - q),  ds = (-rs - dz)/d
(pivot-free LU, x rows first, then the equality rows) against the oracle's contact-space solve (pdipm.py:325-354).
Same iterates in exact arithmetic; this measures what rounding does to the answer, the iteration counts and the best-iterate
choice.   python tools/experiments/primal_numerics.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pdipm_oracle as O
from lcp_physics_amd import scenes

F_GLOBAL = {}

def lu_nopivot_solve(K, r):
    B, n, _ = K.shape
    K = K.clone(); r = r.clone()
    for k in range(n):
        piv = K[:, k, k]
        l = K[:, k + 1:, k] / piv.unsqueeze(1)
        K[:, k + 1:, k + 1:] -= l.unsqueeze(2) * K[:, k, k + 1:].unsqueeze(1)
        r[:, k + 1:] -= l * r[:, k].unsqueeze(1)
    x = torch.zeros_like(r)
    for k in range(n - 1, -1, -1):
        x[:, k] = (r[:, k] - (K[:, k, k + 1:] * x[:, k + 1:]).sum(1)) / K[:, k, k]
    return x

def solve_kkt_primal(k, d, G, A, rx, rs, rz, ry):
    F = F_GLOBAL["F"]; Q = F_GLOBAL["Q"]
    Bsz, m, nz = G.shape
    Dm = 1.0 / d
    if F_GLOBAL.get("delta", 0.0) > 0.0:                       # floor on D = s / z: delta x (effective inverse mass of the row)
        qi = 1.0 / torch.diagonal(Q, dim1=1, dim2=2)
        Dm = torch.maximum(Dm, F_GLOBAL["delta"] * (G * G * qi.unsqueeze(1)).sum(2))
    Mi = F + torch.diag_embed(Dm)
    Bm = torch.linalg.inv(Mi)                      # block diagonal in exact arithmetic; dense inverse here
    q = rs * Dm - rz
    S = Q + G.transpose(1, 2) @ Bm @ G
    rhs_x = -rx + (G.transpose(1, 2) @ (Bm @ q.unsqueeze(2))).squeeze(2)
    neq = k.neq
    if neq > 0:
        K = torch.zeros(Bsz, nz + neq, nz + neq, dtype=Q.dtype)
        K[:, :nz, :nz] = S; K[:, :nz, nz:] = A.transpose(1, 2); K[:, nz:, :nz] = A
        rhs = torch.cat([rhs_x, -ry], 1)
    else:
        K, rhs = S, rhs_x
    sol = lu_nopivot_solve(K, rhs)
    dx = sol[:, :nz]
    dy = sol[:, nz:] if neq > 0 else None
    dz = (Bm @ ((G @ dx.unsqueeze(2)).squeeze(2) - q).unsqueeze(2)).squeeze(2)
    ds = (-rs - dz) / d
    return dx, ds, dz, dy

def run(sc, n_list, label):
    worst = 0.0; it_diff = 0; tot = 0
    for kk in range(sc.B if hasattr(sc, "B") else sc.v.shape[0]):
        one = lambda t: t[kk:kk + 1]
        n = n_list[kk % len(n_list)]
        args = (one(sc.Mdiag), one(sc.v), one(sc.f), sc.dt, sc.c_n[kk:kk + 1, :n], sc.c_p1[kk:kk + 1, :n], sc.c_p2[kk:kk + 1, :n],
                sc.c_i1[kk:kk + 1, :n], sc.c_i2[kk:kk + 1, :n], one(sc.rest), one(sc.fric), one(sc.Je))
        lcp = [None if t is None else t.double() for t in O.assemble_lcp(*args)]
        ref = O.lcp_forward(*lcp)
        F_GLOBAL["Q"], F_GLOBAL["F"] = lcp[0], lcp[6]
        keep = O.solve_kkt
        O.solve_kkt = solve_kkt_primal
        try:
            got = O.lcp_forward(*lcp)
        finally:
            O.solve_kkt = keep
        sc_ = max(1.0, float(ref.x.abs().max()))
        e = float((got.x - ref.x).abs().max()) / sc_
        worst = max(worst, e)
        it_diff += int((got.iters != ref.iters).sum()); tot += 1
        if e > 1e-7:
            print("  scene", kk, "contacts", n, "err", e, "iters", int(ref.iters), int(got.iters), "best resid", float(ref.resid) if hasattr(ref, "resid") else "")
    print(label, "worst scaled |x - x_ref|", worst, " scenes with different iteration count", it_diff, "of", tot)

torch.manual_seed(0)
for dl in (0.0, 1e-9):
    F_GLOBAL["delta"] = dl
    print("---- forward, floor", dl)
    run(scenes.make_stack_scenes(B=16, nbox=11, pts_per_interface=1, seed=71, dtype=torch.float32), [11], "tower of 11, one point per interface")
    run(scenes.make_pile_scenes(B=12, seed=21, dtype=torch.float32), [64, 64, 48, 33, 17, 64, 5, 64, 20, 64, 1, 64], "config 5 piles")
    run(scenes.make_stack_scenes(B=12, nbox=4, pts_per_interface=4, seed=5, dtype=torch.float32), [16], "stacks 4x4")
F_GLOBAL["delta"] = 0.0
sc = scenes.make_pile_scenes(B=24, seed=21, dtype=torch.float32)
run(sc, [64, 64, 48, 33, 17, 64, 5, 64, 20, 64, 1, 64], "config 5 piles")
sc = scenes.make_stack_scenes(B=24, nbox=4, pts_per_interface=4, seed=5, dtype=torch.float32)
run(sc, [16], "config 3 stacks (16 contacts)")


# ---- second experiment: the backward solve (lcp.py:44-50: one KKT solve at the converged iterate, arbitrary right-hand side) with the
# ratios D = s / z floored at DELTA x (the contact's effective inverse mass): where the iteration has converged to machine
# precision D underflows against Q and Q + G^T M^-1 G loses Q
def backward_check(sc, n_list, label, deltas=(0.0, 1e-12, 1e-10, 1e-9, 1e-8)):
    worst = {d: 0.0 for d in deltas}
    for kk in range(sc.v.shape[0]):
        one = lambda t: t[kk:kk + 1]
        n = n_list[kk % len(n_list)]
        if n == 0:
            continue
        args = (one(sc.Mdiag), one(sc.v), one(sc.f), sc.dt, sc.c_n[kk:kk + 1, :n], sc.c_p1[kk:kk + 1, :n], sc.c_p2[kk:kk + 1, :n],
                sc.c_i1[kk:kk + 1, :n], sc.c_i2[kk:kk + 1, :n], one(sc.rest), one(sc.fric), one(sc.Je))
        lcp = [None if t is None else t.double() for t in O.assemble_lcp(*args)]
        Q, p, G, h, A, b, F = lcp
        ref = O.lcp_forward(*lcp)
        d = ref.z / ref.s
        k = O.pre_factor_kkt(Q, G, F, A)
        O.factor_kkt(k, d)
        g = torch.randn(1, Q.shape[1], dtype=torch.float64, generator=torch.Generator().manual_seed(kk))
        zm = torch.zeros_like(d)
        ze = torch.zeros(1, A.shape[1], dtype=torch.float64)
        dx_ref = O.solve_kkt(k, d, G, A, g, zm, zm, ze)[0]
        F_GLOBAL["Q"], F_GLOBAL["F"] = Q, F
        qi = 1.0 / torch.diagonal(Q, dim1=1, dim2=2)
        w = (G * G * qi.unsqueeze(1)).sum(2)                       # effective inverse mass of every row (0 for the cone rows)
        for dl in deltas:
            F_GLOBAL["delta"] = dl
            dx = solve_kkt_primal(k, d, G, A, g, zm, zm, ze)[0]
            F_GLOBAL["delta"] = 0.0
            e = float((dx - dx_ref).abs().max() / dx_ref.abs().max())
            worst[dl] = max(worst[dl], e)
    print(label, "backward solve, worst relative |dx - dx_dual| per floor:", {k: "%.1e" % v for k, v in worst.items()})

backward_check(scenes.make_pile_scenes(B=12, seed=21, dtype=torch.float32), [64, 64, 48, 33, 17, 64, 5, 64, 20, 64, 1, 64], "config 5 piles")
backward_check(scenes.make_stack_scenes(B=12, nbox=4, pts_per_interface=4, seed=5, dtype=torch.float32), [16], "stacks 4x4")
backward_check(scenes.make_stack_scenes(B=16, nbox=11, pts_per_interface=1, seed=71, dtype=torch.float32), [11], "tower of 11, one point per interface")
backward_check(scenes.make_stack_scenes(B=16, nbox=6, pts_per_interface=4, seed=66, dtype=torch.float32), [24], "stack of 6, 24 contacts")


# ---- third experiment: the same backward solve with the floored body-space factorisation as the solver of an iterative refinement on
# the UNREDUCED step equations (residuals evaluated with M, not M^-1: nothing large is multiplied by anything cancelled)
def refine_check(sc, n_list, label, delta=1e-9, steps=(0, 1, 2)):
    worst = {s_: 0.0 for s_ in steps}
    for kk in range(sc.v.shape[0]):
        one = lambda t: t[kk:kk + 1]
        n = n_list[kk % len(n_list)]
        if n == 0:
            continue
        args = (one(sc.Mdiag), one(sc.v), one(sc.f), sc.dt, sc.c_n[kk:kk + 1, :n], sc.c_p1[kk:kk + 1, :n], sc.c_p2[kk:kk + 1, :n],
                sc.c_i1[kk:kk + 1, :n], sc.c_i2[kk:kk + 1, :n], one(sc.rest), one(sc.fric), one(sc.Je))
        lcp = [None if t is None else t.double() for t in O.assemble_lcp(*args)]
        Q, p, G, h, A, b, F = lcp
        ref = O.lcp_forward(*lcp)
        d = ref.z / ref.s
        k = O.pre_factor_kkt(Q, G, F, A)
        O.factor_kkt(k, d)
        g = torch.randn(1, Q.shape[1], dtype=torch.float64, generator=torch.Generator().manual_seed(kk))
        zm = torch.zeros_like(d)
        ze = torch.zeros(1, A.shape[1], dtype=torch.float64)
        dx_ref, _, dz_ref, dy_ref = O.solve_kkt(k, d, G, A, g, zm, zm, ze)
        F_GLOBAL["Q"], F_GLOBAL["F"], F_GLOBAL["delta"] = Q, F, delta
        M = F + torch.diag_embed(1.0 / d)
        mv = lambda Mx, v: (Mx @ v.unsqueeze(2)).squeeze(2)
        dx, _, dz, dy = solve_kkt_primal(k, d, G, A, g, zm, zm, ze)
        for it in range(max(steps) + 1):
            if it in worst:
                worst[it] = max(worst[it], float((dx - dx_ref).abs().max() / max(1e-300, float(g.abs().max() / torch.diagonal(Q, dim1=1, dim2=2).min()))))
            # residuals of  Q dx + G^T dz + A^T dy = -g ,  G dx - M dz = 0 ,  A dx = 0
            r1 = -g - (mv(Q, dx) + mv(G.transpose(1, 2), dz) + mv(A.transpose(1, 2), dy))
            r3 = -(mv(G, dx) - mv(M, dz))
            r2 = -mv(A, dx)
            ddx, _, ddz, ddy = solve_kkt_primal(k, d, G, A, -r1, zm, -r3, -r2)
            dx, dz, dy = dx + ddx, dz + ddz, dy + ddy
        F_GLOBAL["delta"] = 0.0
    print(label, "floor %.0e, error of dx (scaled by |g| / min Q) after 0 / 1 / 2 refinement steps:" % delta, {k: "%.1e" % v for k, v in worst.items()})

for dl in (1e-9, 1e-7):
    refine_check(scenes.make_pile_scenes(B=12, seed=21, dtype=torch.float32), [64, 64, 48, 33, 17, 64, 5, 64, 20, 64, 1, 64], "config 5 piles", dl)
    refine_check(scenes.make_stack_scenes(B=12, nbox=4, pts_per_interface=4, seed=5, dtype=torch.float32), [16], "stacks 4x4", dl)
    refine_check(scenes.make_stack_scenes(B=16, nbox=11, pts_per_interface=1, seed=71, dtype=torch.float32), [11], "tower of 11", dl)
    refine_check(scenes.make_stack_scenes(B=16, nbox=6, pts_per_interface=4, seed=66, dtype=torch.float32), [24], "stack of 6", dl)

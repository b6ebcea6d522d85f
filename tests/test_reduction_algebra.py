"""CPU: the contact-structured reduction used by the wave64 kernels (lcp_wave64.hip, `Red`) is an EXACT
reformulation of the T-solve of pdipm.py:325-354 / :414-454, and is backward stable without pivoting.

For G = [Jc; Jf; 0] with +/- paired friction rows and F = [[0,0,0],[0,0,E],[mu,-E^T,0]]
(physics/engines.py:67-73) the 4nc x 4nc system (R + diag(s/z)) dz = r reduces to 2nc unknowns
(a = normal multipliers, u = b1 - b2) after eliminating (w = b1 + b2, g = gamma) per contact."""
import torch

from lcp_physics_amd import scenes
from oracle import pdipm_oracle as O


def reduced_solve(R, s, z, mu, r, nc, pivot):
    D = s / z
    n = slice(0, nc)
    f1 = torch.arange(nc) * 2 + nc
    f2 = f1 + 1
    g = slice(3 * nc, 4 * nc)
    Dn, D1, D2, Dg = D[:, n], D[:, f1], D[:, f2], D[:, g]
    Wnn, Wnt = R[:, n, n], R[:, n][:, :, f1]
    Wtn, Wtt = R[:, f1][:, :, n], R[:, f1][:, :, f1]
    rn, r1, r2, rg = r[:, n], r[:, f1], r[:, f2], r[:, g]
    Sp, Sm = 0.5 * (D1 + D2), 0.5 * (D1 - D2)
    idet = 1.0 / (Sp * Dg + 2)
    w0 = (Dg * (r1 + r2) - 2 * rg) * idet
    wa, wu = 2 * mu * idet, -Dg * Sm * idet
    M = torch.zeros(R.shape[0], 2 * nc, 2 * nc, dtype=R.dtype)
    M[:, :nc, :nc] = Wnn + torch.diag_embed(Dn)
    M[:, :nc, nc:] = Wnt
    M[:, nc:, :nc] = Wtn + torch.diag_embed(0.5 * Sm * wa)
    M[:, nc:, nc:] = Wtt + torch.diag_embed(0.5 * (Sp + Sm * wu))
    rhs = torch.cat([rn, 0.5 * (r1 - r2) - 0.5 * Sm * w0], 1)
    if pivot:
        sol = torch.linalg.solve(M, rhs.unsqueeze(-1)).squeeze(-1)
    else:
        LU, piv = O._lu_nopivot(M)
        sol = torch.linalg.lu_solve(LU, piv, rhs.unsqueeze(-1)).squeeze(-1)
    a, u = sol[:, :nc], sol[:, nc:]
    w = w0 + wa * a + wu * u
    gg = ((r1 + r2) - Sm * u + Sp * (rg - mu * a)) * idet
    dz = torch.zeros_like(r)
    dz[:, n], dz[:, f1], dz[:, f2], dz[:, g] = a, 0.5 * (w + u), 0.5 * (w - u), gg
    return dz


def test_reduced_system_is_exact_and_stable_without_pivoting():
    sc = scenes.make_stack_scenes(B=48, nbox=4, pts_per_interface=4, seed=1236, dtype=torch.float32)
    lcp = [None if t is None else t.double() for t in O.assemble_lcp(*sc.assembly_args())]
    Q, p, G, h, A, b, F = lcp
    B, m, _ = G.shape
    nc = m // 4
    R = O.pre_factor_kkt(Q, G, F, A).R
    mu = F[:, 3 * nc:, :nc].diagonal(dim1=1, dim2=2)
    trace = []
    O.lcp_forward(*lcp, trace=trace)
    for it, st in enumerate(trace):
        s, z = st["s"], st["z"]
        T = R + torch.diag_embed(s / z)
        r = torch.randn(B, m, generator=torch.Generator().manual_seed(it), dtype=torch.float64)
        for pivot in (True, False):
            dz = reduced_solve(R, s, z, mu, r, nc, pivot)
            resid = (torch.bmm(T, dz.unsqueeze(-1)).squeeze(-1) - r).norm(dim=1)
            berr = resid / (T.norm(dim=(1, 2)) * dz.norm(dim=1) + r.norm(dim=1))
            assert float(berr.max()) < 1e-14, (it, pivot, float(berr.max()))     # even at cond(T) ~ 1e19
        if it <= 3:                                                              # while T is well conditioned
            ref = torch.linalg.solve(T, r.unsqueeze(-1)).squeeze(-1)
            assert float(((dz - ref).norm(dim=1) / ref.norm(dim=1)).max()) < 1e-9

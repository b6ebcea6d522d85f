"""CPU oracle for the batched PDIPM LCP contact solve.  TEST INFRASTRUCTURE ONLY.

This is a clean-room restatement (torch, CPU, fp64 by default) of the reference
algorithm, used as the *checker* for the HIP path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product package never does.

Parity status: PINNED against the reference's own code.  The reference has no
golden vectors, so `oracle/make_golden.py` runs the unmodified
reference (through `oracle/ref_shim.py`) on scenes built with its own `World`, and
`tests/test_oracle_golden.py` checks this restatement against those fixtures
(`tests/golden/*.npz`) and, when `/root/reference` is present, against the
reference live.

Reference lines followed (all under /root/reference/lcp_physics):
  lcp/solvers/pdipm.py:49-179   forward (the PDIPM loop)        -> `pdipm_forward`
  lcp/solvers/pdipm.py:182-186  get_step                        -> `get_step`
  lcp/solvers/pdipm.py:325-354  solve_kkt                       -> `solve_kkt`
  lcp/solvers/pdipm.py:357-408  pre_factor_kkt                  -> `pre_factor_kkt`
  lcp/solvers/pdipm.py:414-454  factor_kkt                      -> `factor_kkt`
  lcp/lcp.py:22-35, 37-64       LCPFunction forward / backward  -> `lcp_forward`, `lcp_backward`
  physics/engines.py:26-78      PdipmEngine.solve_dynamics      -> `assemble_lcp`, `solve_dynamics`
  physics/world.py:144-234      restitutions, Jc, Jf, mu, E     -> `contact_jacobians`, ...
  physics/bodies.py:80-82       Body.move                       -> `integrate`

Batch semantics: the reference couples batch elements in two places (termination
is batch-global, `pdipm.py:116-119,133`; `get_step` fills with a batch-global max,
`pdipm.py:184`).  `physics` only ever calls it with batch 1, so the semantics that
matter - and the ones the per-scene HIP kernels implement - are "each scene solved
as if it were a batch of one".  This oracle is vectorised over the batch but keeps
per-scene state (done flags, per-row max), i.e. batch-1 semantics for every scene.
"""
import torch

# ----------------------------------------------------------------------------
# small dense helpers
# ----------------------------------------------------------------------------


def _lu_nopivot(M):
    """Doolittle LU without pivoting, batched (mirror of a pivot-free device LU).
    Returns (LU, pivots) with identity 1-based pivots so that lu_solve works."""
    LU = M.clone()
    n = M.shape[-1]
    for k in range(n - 1):
        LU[:, k + 1:, k] = LU[:, k + 1:, k] / LU[:, k:k + 1, k]
        LU[:, k + 1:, k + 1:] -= LU[:, k + 1:, k:k + 1] * LU[:, k:k + 1, k + 1:]
    piv = torch.arange(1, n + 1, dtype=torch.int32).unsqueeze(0).repeat(M.shape[0], 1)
    return LU, piv


def lu_factor(M, pivot=True):
    """pdipm.py:15-28 `btrifact_hack`: partial pivoting on CPU, none on GPU."""
    if pivot:
        LU, piv, _info = torch.linalg.lu_factor_ex(M)
        return LU, piv
    return _lu_nopivot(M)


def lu_solve(LU, v):
    """`Tensor.btrisolve` for a [B,n] right-hand side."""
    return torch.linalg.lu_solve(LU[0], LU[1], v.unsqueeze(-1)).squeeze(-1)


def _mv(M, v):
    return torch.bmm(M, v.unsqueeze(-1)).squeeze(-1)


def _mtv(M, v):
    return torch.bmm(M.transpose(1, 2), v.unsqueeze(-1)).squeeze(-1)


def _nanmax_rows(a):
    """Row max that propagates NaN like `Tensor.max()` does."""
    return a.max(dim=1)[0]


def _nanmin_rows(a):
    return a.min(dim=1)[0]


# ----------------------------------------------------------------------------
# KKT machinery
# ----------------------------------------------------------------------------


class KKT:
    """Cached one-time factorisations (`pdipm.py:357-408`).

    The reference stores a partially completed block-LU of
        S = [[A Q^-1 A^T, A Q^-1 G^T], [G Q^-1 A^T, G Q^-1 G^T + F + D^-1]]
    (`S_LU`) plus `R`.  Mathematically that is: an LU of S11 = A Q^-1 A^T, the
    coupling block GA = G Q^-1 A^T, and the Schur complement
        R = G Q^-1 G^T + F - GA S11^-1 GA^T            (pdipm.py:378-403)
    whose diagonal is completed by `factor_kkt` once d is known.
    """

    def __init__(self):
        self.Q_LU = None
        self.S11_LU = None
        self.GA = None
        self.R = None
        self.T_LU = None
        self.neq = 0
        self.pivot = True


def pre_factor_kkt(Q, G, F, A, pivot=True):
    """pdipm.py:357-408."""
    k = KKT()
    k.pivot = pivot
    k.neq = A.shape[1] if A is not None and A.dim() == 3 else 0
    k.Q_LU = lu_factor(Q, pivot=True)                                  # :362
    invQ_GT = torch.linalg.lu_solve(k.Q_LU[0], k.Q_LU[1], G.transpose(1, 2))
    R = torch.bmm(G, invQ_GT) + F                                      # :378-379
    if k.neq > 0:
        invQ_AT = torch.linalg.lu_solve(k.Q_LU[0], k.Q_LU[1], A.transpose(1, 2))   # :383
        S11 = torch.bmm(A, invQ_AT)                                    # :384
        k.GA = torch.bmm(G, invQ_AT)                                   # :385
        k.S11_LU = lu_factor(S11, pivot=True)                          # :387
        Tm = torch.linalg.lu_solve(k.S11_LU[0], k.S11_LU[1], k.GA.transpose(1, 2))  # :395
        R = R - torch.bmm(k.GA, Tm)                                    # :403
    k.R = R
    return k


def factor_kkt(k, d):
    """pdipm.py:414-454: T = R + diag(1/d), LU(T)."""
    T = k.R.clone()
    idx = torch.arange(T.shape[1])
    T[:, idx, idx] += 1.0 / d
    k.T_LU = lu_factor(T, pivot=k.pivot)
    # exactly singular U (a zero pivot): the reference's bare `except` around factor_kkt
    # (pdipm.py:99-102) returns the best iterate; report it so the caller can do the same.
    dg = torch.diagonal(k.T_LU[0], dim1=1, dim2=2)
    return (dg == 0).any(dim=1)


def _solve_S(k, hy, hz):
    """Solve S [wy; wz] = [hy; hz] by block elimination of the equality block -
    what `h.btrisolve(*S_LU)` does with the assembled block LU (pdipm.py:345)."""
    if k.neq > 0:
        t = lu_solve(k.S11_LU, hy)
        r2 = hz - _mv(k.GA, t)
        wz = lu_solve(k.T_LU, r2)
        wy = lu_solve(k.S11_LU, hy - _mtv(k.GA, wz))
        return wy, wz
    return None, lu_solve(k.T_LU, hz)


def solve_kkt(k, d, G, A, rx, rs, rz, ry):
    """pdipm.py:325-354.  Solves K [dx;ds;dz;dy] = -[rx;rs;rz;ry]."""
    invQ_rx = lu_solve(k.Q_LU, rx)                                     # :333
    hz = _mv(G, invQ_rx) + rs / d - rz                                 # :337-340
    hy = (_mv(A, invQ_rx) - ry) if k.neq > 0 else None
    wy, wz = _solve_S(k, hy, hz)
    wz = -wz                                                           # :342
    g1 = -rx - _mtv(G, wz)                                             # :344
    if k.neq > 0:
        wy = -wy
        g1 = g1 - _mtv(A, wy)                                          # :346
    g2 = -rs - wz                                                      # :347
    dx = lu_solve(k.Q_LU, g1)                                          # :349
    ds = g2 / d                                                        # :350
    return dx, ds, wz, wy


def get_step(v, dv):
    """pdipm.py:182-186 with the batch-global max read per scene (batch-1 semantics).

    a = -v/dv; entries with dv > 0 are replaced by max(1.0, max(a)); NaN in `a`
    makes Python's max(1.0, nan) return 1.0; the row min then propagates NaN.
    """
    a = -v / dv
    amax = _nanmax_rows(a)
    fill = torch.where(amax > 1.0, amax, torch.ones_like(amax))
    a = torch.where(dv > 0, fill.unsqueeze(1).expand_as(a), a)
    return _nanmin_rows(a)


def pdipm_forward(Q, p, G, h, A, b, F, k, eps=1e-12, not_improved_lim=3, max_iter=10,
                  trace=None):
    """pdipm.py:49-179, per-scene semantics.  Returns best (x, y, z, s) plus the number
    of loop iterations each scene executed (`iters`: factorisations inside the loop)."""
    B, m, nz = G.shape
    neq = k.neq
    dt = Q.dtype
    one = torch.ones(B, dtype=dt)

    d = torch.ones(B, m, dtype=dt)
    factor_kkt(k, d)                                                    # :58-59
    x, s, z, y = solve_kkt(k, d, G, A, p, torch.zeros(B, m, dtype=dt), -h,
                           -b if neq > 0 else None)                     # :60-63
    smin = _nanmin_rows(s)                                              # :66-69
    s = torch.where((smin <= 0).unsqueeze(1), s - smin.unsqueeze(1) + 1, s)
    zmin = _nanmin_rows(z)                                              # :72-75
    z = torch.where((zmin <= 0).unsqueeze(1), z - zmin.unsqueeze(1) + 1, z)

    best_r = torch.full((B,), float("inf"), dtype=dt)
    bx, bs, bz = x.clone(), s.clone(), z.clone()
    by = y.clone() if neq > 0 else None
    have_best = torch.zeros(B, dtype=torch.bool)
    n_not = torch.zeros(B, dtype=torch.int64)
    done = torch.zeros(B, dtype=torch.bool)
    iters = torch.zeros(B, dtype=torch.int32)

    for it in range(max_iter):                                          # :80
        rx = _mtv(G, z) + _mv(Q, x) + p                                 # :82-85
        if neq > 0:
            rx = rx + _mtv(A, y)
        rs = z                                                          # :86
        rz = _mv(G, x) + s - h - _mv(F, z)                              # :87-88
        ry = (_mv(A, x) - b) if neq > 0 else None                       # :89-90
        mu = torch.abs((s * z).sum(1) / m)                              # :91
        resid = rz.norm(dim=1) + rx.norm(dim=1) + m * mu                # :92-96
        if neq > 0:
            resid = resid + ry.norm(dim=1)
        d = z / s                                                       # :98
        singular = factor_kkt(k, d)                                     # :99-102
        iters = iters + (~done).to(torch.int32)
        if it > 0:
            done = done | singular                                      # except: return best

        improved = resid < best_r                                       # :115 (NaN -> False)
        first = ~have_best
        take = (first | improved) & ~done                               # :107-128
        n_not = torch.where(first | improved, torch.zeros_like(n_not), n_not + 1)
        best_r = torch.where(take, resid, best_r)
        bx = torch.where(take.unsqueeze(1), x, bx)
        bs = torch.where(take.unsqueeze(1), s, bs)
        bz = torch.where(take.unsqueeze(1), z, bz)
        if neq > 0:
            by = torch.where(take.unsqueeze(1), y, by)
        have_best = have_best | ~done
        if trace is not None:
            trace.append({"it": it, "resid": resid.clone(), "mu": mu.clone(),
                          "done": done.clone(), "x": x.clone(), "s": s.clone(), "z": z.clone()})
        stop = (n_not == not_improved_lim) | (best_r < eps) | (mu > 1e100)   # :133
        done = done | stop
        if bool(done.all()):
            break

        dx_a, ds_a, dz_a, dy_a = solve_kkt(k, d, G, A, rx, rs, rz, ry)   # :138-139
        alpha = torch.min(torch.min(get_step(z, dz_a), get_step(s, ds_a)), one)  # :142-144
        t1 = s + alpha.unsqueeze(1) * ds_a                              # :146
        t2 = z + alpha.unsqueeze(1) * dz_a
        sig = ((t1 * t2).sum(1) / (s * z).sum(1)) ** 3                  # :148-150
        rs_c = ((-mu * sig).unsqueeze(1) + ds_a * dz_a) / s             # :153
        zx = torch.zeros(B, nz, dtype=dt)
        zm = torch.zeros(B, m, dtype=dt)
        ze = torch.zeros(B, neq, dtype=dt) if neq > 0 else None
        dx_c, ds_c, dz_c, dy_c = solve_kkt(k, d, G, A, zx, rs_c, zm, ze)  # :157-158
        dx, ds, dz = dx_a + dx_c, ds_a + ds_c, dz_a + dz_c              # :160-163
        alpha = torch.min(0.999 * torch.min(get_step(z, dz), get_step(s, ds)), one)  # :164-166
        upd = (~done).unsqueeze(1)
        x = torch.where(upd, x + alpha.unsqueeze(1) * dx, x)           # :171-174
        s = torch.where(upd, s + alpha.unsqueeze(1) * ds, s)
        z = torch.where(upd, z + alpha.unsqueeze(1) * dz, z)
        if neq > 0:
            y = torch.where(upd, y + alpha.unsqueeze(1) * (dy_a + dy_c), y)

    return bx, by, bz, bs, iters, best_r


# ----------------------------------------------------------------------------
# the differentiable op  (lcp/lcp.py)
# ----------------------------------------------------------------------------


class Solution:
    pass


def lcp_forward(Q, p, G, h, A, b, F, eps=1e-12, not_improved_lim=3, max_iter=10, pivot=True,
                trace=None):
    """lcp.py:22-35.  A may be None / empty for "no equalities" (lcp.py:24)."""
    if A is not None and A.dim() != 3:
        A, b = None, None
    k = pre_factor_kkt(Q, G, F, A, pivot=pivot)
    x, y, z, s, iters, resid = pdipm_forward(Q, p, G, h, A, b, F, k, eps=eps,
                                             not_improved_lim=not_improved_lim,
                                             max_iter=max_iter, trace=trace)
    sol = Solution()
    sol.x, sol.y, sol.z, sol.s, sol.iters, sol.resid, sol.kkt = x, y, z, s, iters, resid, k
    return sol


def _outer(a, b_):
    return a.unsqueeze(2) * b_.unsqueeze(1)


def lcp_backward(sol, Q, p, G, h, A, b, F, dl_dx, adjoint=False):
    """lcp.py:37-64.  The reference solves with K (not K^T) - exact only for symmetric
    F.  `adjoint=True` gives the transposed-system variant instead
    (never the parity target)."""
    if A is not None and A.dim() != 3:
        A = None
    k = sol.kkt
    B, m, nz = G.shape
    d = sol.z / sol.s                                                   # :44
    dt = Q.dtype
    if not adjoint:
        factor_kkt(k, d)                                                # :46
        dx, _, dlam, dnu = solve_kkt(k, d, G, A, dl_dx, torch.zeros(B, m, dtype=dt),
                                     torch.zeros(B, m, dtype=dt),
                                     torch.zeros(B, k.neq, dtype=dt) if k.neq else None)  # :47-50
    else:
        e = k.neq
        n = nz + 2 * m + e
        K = torch.zeros(B, n, n, dtype=dt)
        K[:, :nz, :nz] = Q
        K[:, :nz, nz + m:nz + 2 * m] = G.transpose(1, 2)
        ar = torch.arange(m)
        K[:, nz + ar, nz + ar] = d
        K[:, nz + ar, nz + m + ar] = 1
        K[:, nz + m:nz + 2 * m, :nz] = G
        K[:, nz + m + ar, nz + ar] = 1
        K[:, nz + m:nz + 2 * m, nz + m:nz + 2 * m] = -F
        if e:
            K[:, :nz, nz + 2 * m:] = A.transpose(1, 2)
            K[:, nz + 2 * m:, :nz] = A
        rhs = torch.zeros(B, n, dtype=dt)
        rhs[:, :nz] = -dl_dx
        sol_ = torch.linalg.solve(K.transpose(1, 2), rhs)
        dx, dlam = sol_[:, :nz], sol_[:, nz + m:nz + 2 * m]
        dnu = sol_[:, nz + 2 * m:] if e else None
    g = {}
    g["dp"] = dx                                                        # :52
    g["dG"] = _outer(dlam, sol.x) + _outer(sol.z, dx)                   # :53
    g["dF"] = -_outer(dlam, sol.z)                                      # :54
    g["dh"] = -dlam                                                     # :55
    if k.neq > 0:
        g["dA"] = _outer(dnu, sol.x) + _outer(sol.y, dx)                # :57
        g["db"] = -dnu                                                  # :58
    else:
        g["dA"], g["db"] = None, None
    g["dQ"] = 0.5 * (_outer(dx, sol.x) + _outer(sol.x, dx))             # :61
    return g


def dense_kkt_matrix(Q, G, A, F, d):
    """K = [[Q,0,G^T,A^T],[0,D,I,0],[G,I,-F,0],[A,0,0,0]] (pdipm.py:325-354)."""
    B, m, nz = G.shape
    e = A.shape[1] if A is not None and A.dim() == 3 else 0
    n = nz + 2 * m + e
    K = torch.zeros(B, n, n, dtype=Q.dtype)
    ar = torch.arange(m)
    K[:, :nz, :nz] = Q
    K[:, :nz, nz + m:nz + 2 * m] = G.transpose(1, 2)
    K[:, nz + ar, nz + ar] = d
    K[:, nz + ar, nz + m + ar] = 1
    K[:, nz + m:nz + 2 * m, :nz] = G
    K[:, nz + m + ar, nz + ar] = 1
    K[:, nz + m:nz + 2 * m, nz + m:nz + 2 * m] = -F
    if e:
        K[:, :nz, nz + 2 * m:] = A.transpose(1, 2)
        K[:, nz + 2 * m:, :nz] = A
    return K


# ----------------------------------------------------------------------------
# physics: contact Jacobians, LCP assembly, integrator
# ----------------------------------------------------------------------------


def cross_2d(a, b_):
    """physics/utils.py:93-96."""
    return a[..., 0] * b_[..., 1] - a[..., 1] * b_[..., 0]


def left_orthogonal(v):
    """physics/utils.py:99-102."""
    return torch.stack([v[..., 1], -v[..., 0]], dim=-1)


def contact_jacobians(n, p1, p2, i1, i2, nb):
    """world.py:172-211 (`Jc`, `Jf` with 2 friction directions), batched.

    n, p1, p2: [B,nc,2]; i1, i2: [B,nc] integer body indices; returns
    Jc [B,nc,3nb], Jf [B,2nc,3nb]."""
    B, nc, _ = n.shape
    dt = n.dtype
    nz = 3 * nb
    Jc = torch.zeros(B, nc, nz, dtype=dt)
    Jf = torch.zeros(B, 2 * nc, nz, dtype=dt)
    t1 = left_orthogonal(n)                                             # world.py:191
    bi = torch.arange(B).unsqueeze(1).expand(B, nc)
    ci = torch.arange(nc).unsqueeze(0).expand(B, nc)
    i1 = i1.long()
    i2 = i2.long()
    # normal rows (world.py:177-183)
    j1 = torch.stack([cross_2d(p1, n), n[..., 0], n[..., 1]], dim=-1)
    j2 = -torch.stack([cross_2d(p2, n), n[..., 0], n[..., 1]], dim=-1)
    # friction rows (world.py:196-210): rows 2i (dir1) and 2i+1 (dir2 = -dir1)
    f1 = torch.stack([cross_2d(p1, t1), t1[..., 0], t1[..., 1]], dim=-1)
    f2 = torch.stack([cross_2d(p2, t1), t1[..., 0], t1[..., 1]], dim=-1)
    for c in range(3):
        # body 2 is written after body 1 in the reference (plain assignment), keep order
        Jc[bi, ci, 3 * i1 + c] = j1[..., c]
        Jc[bi, ci, 3 * i2 + c] = j2[..., c]
        Jf[bi, 2 * ci, 3 * i1 + c] = f1[..., c]
        Jf[bi, 2 * ci + 1, 3 * i1 + c] = -f1[..., c]
        Jf[bi, 2 * ci, 3 * i2 + c] = -f2[..., c]
        Jf[bi, 2 * ci + 1, 3 * i2 + c] = f2[..., c]
    return Jc, Jf


def contact_coefficients(rest, fric, i1, i2):
    """world.py:144-151 (restitutions) and :213-224 (mu): arithmetic means."""
    i1 = i1.long()
    i2 = i2.long()
    r = 0.5 * (torch.gather(rest, 1, i1) + torch.gather(rest, 1, i2))
    mu = 0.5 * (torch.gather(fric, 1, i1) + torch.gather(fric, 1, i2))
    return r, mu


def assemble_lcp(Mdiag, v, f, dt, n, p1, p2, i1, i2, rest, fric, Je):
    """engines.py:31-32, 50-74: build (Q, p, G, h, A, b, F) for the contact branch.

    Mdiag, v, f: [B,nb,3]; Je: [B,e,nz] or None.  Returns dense tensors."""
    B, nb, _ = v.shape
    nc = n.shape[1]
    nz = 3 * nb
    dtp = v.dtype
    Md = Mdiag.reshape(B, nz)
    vv = v.reshape(B, nz)
    u = Md * vv + dt * f.reshape(B, nz)                                 # engines.py:32
    Jc, Jf = contact_jacobians(n, p1, p2, i1, i2, nb)
    r, mu = contact_coefficients(rest, fric, i1, i2)
    hv = _mv(Jc, vv) * r                                                # engines.py:53
    m = 4 * nc
    G = torch.zeros(B, m, nz, dtype=dtp)
    G[:, :nc] = Jc
    G[:, nc:3 * nc] = Jf                                                # :67-68
    Fm = torch.zeros(B, m, m, dtype=dtp)
    ci = torch.arange(nc)
    Fm[:, nc + 2 * ci, 3 * nc + ci] = 1                                 # :70  E
    Fm[:, nc + 2 * ci + 1, 3 * nc + ci] = 1
    Fm[:, 3 * nc + ci, ci] = mu                                         # :71  mu
    Fm[:, 3 * nc + ci, nc + 2 * ci] = -1                                # :72-73  -E^T
    Fm[:, 3 * nc + ci, nc + 2 * ci + 1] = -1
    h = torch.zeros(B, m, dtype=dtp)
    h[:, :nc] = hv                                                      # :74
    Q = torch.diag_embed(Md)
    if Je is not None and Je.numel() > 0:
        A = Je
        b = torch.zeros(B, Je.shape[1], dtype=dtp)
    else:
        A, b = None, None
    return Q, u, G, h, A, b, Fm


def solve_dynamics(Mdiag, v, f, dt, n, p1, p2, i1, i2, rest, fric, Je, **kw):
    """engines.py:50-78 contact branch: new_v = -x."""
    Q, u, G, h, A, b, Fm = assemble_lcp(Mdiag, v, f, dt, n, p1, p2, i1, i2, rest, fric, Je)
    sol = lcp_forward(Q, u, G, h, A, b, Fm, **kw)
    B, nb, _ = v.shape
    return (-sol.x).reshape(B, nb, 3), sol, (Q, u, G, h, A, b, Fm)


def assemble_post_stabilization(Mdiag, v, n, p1, p2, i1, i2, rest, Je):
    """engines.py:80-116, contact case: the frictionless LCP (Q = M, p = 0, G = Jc, h = gc, A = Je, b = ge, F = 0)
    with gc = Jc v + Jc v * -restitutions (:87-89) and ge = Je v (:86)."""
    B, nb, _ = v.shape
    nc = n.shape[1]
    nz = 3 * nb
    dtp = v.dtype
    vv = v.reshape(B, nz)
    Jc, _ = contact_jacobians(n, p1, p2, i1, i2, nb)
    r, _ = contact_coefficients(rest, rest, i1, i2)
    jv = _mv(Jc, vv)
    gc = jv + jv * -r
    Q = torch.diag_embed(Mdiag.reshape(B, nz))
    if Je is not None and Je.numel() > 0:
        A, b = Je, _mv(Je, vv)
    else:
        A, b = None, None
    return Q, torch.zeros(B, nz, dtype=dtp), Jc, gc, A, b, torch.zeros(B, nc, nc, dtype=dtp)


def post_stabilization(Mdiag, v, n, p1, p2, i1, i2, rest, Je, **kw):
    """engines.py:80-116 contact case: dp = -x (:115)."""
    lcp = assemble_post_stabilization(Mdiag, v, n, p1, p2, i1, i2, rest, Je)
    sol = lcp_forward(*lcp, **kw)
    B, nb, _ = v.shape
    return (-sol.x).reshape(B, nb, 3), sol, lcp


def integrate(p, v_new, dt):
    """bodies.py:80-82: p <- p + v*dt."""
    return p + v_new * dt

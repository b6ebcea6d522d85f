"""Algorithmic FLOP / byte counts of one scene-step - the figures the roofline
fraction in bench.py is computed from.  Dense general-Q formulation exactly as the reference
performs it; a multiply-add counts 2; `it` = PDIPM loop iterations executed by that scene."""


def flops_forward(nz, m, e, it):
    k = e + m
    P = (2.0 / 3) * nz ** 3 + 2 * nz * nz * m + 2 * m * m * nz + m * m
    if e > 0:
        P += 2 * nz * nz * e + 2 * e * e * nz + 2 * m * nz * e + (2.0 / 3) * e ** 3 + 2 * e ** 3 + 6 * m * e * e + 2 * m * m * e
    Fk = (2.0 / 3) * m ** 3 + m
    Sk = 4 * nz * nz + 4 * m * nz + 4 * e * nz + 2 * k * k + 6 * m
    Rk = 2 * nz * nz + 4 * m * nz + 4 * e * nz + 2 * m * m + 10 * m
    I = Fk + 2 * Sk + Rk + 20 * m
    return P + Fk + Sk + it * I


def flops_forward_executed(nz, nc, e, it):
    """FLOPs the contact-structured kernels really EXECUTE per scene (lcp_quad.hip / lcp_big.hip): the 4nc inequality
    rows are reduced exactly to n = 2nc unknowns (normal multipliers + friction differences), Q is diagonal, and the
    corrector solve skips the products of its zero right-hand sides.  Dense arithmetic on the reduced system is counted
    (the kernels do not exploit the sparsity of J inside W = J P J^T), element-wise work as a small multiple of nc."""
    n = 2 * nc
    P = 2 * n * n * nz + (2 * n * nz * e + 2 * e * e * nz + 2 * n * e * e + 2 * n * n * e if e > 0 else 0)   # W = J P J^T (+ equality correction)
    Fk = (2.0 / 3) * n ** 3 + 12 * nc                   # LU of the reduced matrix + the per-contact 2x2 eliminations
    prod = 2 * n * nz                                   # one J v or J^T w product
    tri = 2 * n * n                                     # the two triangular sweeps
    eq = (4 * e * nz + 4 * n * e + 2 * e * e) if e > 0 else 0
    S_full = 2 * prod + tri + eq + 40 * nc + 2 * nz     # solve_kkt with a full right-hand side
    S_corr = prod + tri + (2 * n * e + 2 * e * e + 2 * e * nz if e > 0 else 0) + 40 * nc + 2 * nz   # rx = ry = 0
    Rk = 2 * prod + (4 * e * nz if e > 0 else 0) + 2 * nz + 30 * nc
    I = Fk + S_full + S_corr + Rk + 60 * nc
    return P + Fk + S_full + it * I


SIZED_SHAPES = {(15, 3), (9, 3), (12, 3), (6, 3)}       # (nz, neq) with compile-time instantiations (lcp_quad_n*e*.hip): columns trimmed


def flops_forward_executed_body_space(nz, nc, e, it, pinned=True, trimmed=None):
    """FLOPs the body-space variant executes per scene (lcp_quad.hip ALG = 1 / 2, lcp_primal.hip).  Per factorisation: the
    formation of Q + G^T M^-1 G (per contact the two rows P = B [jc; jt] and two rank-1 updates of the matrix block, plus the
    closed-form 2 x 2 block B of the contact, ~30) and the LU: `pinned` (ALG = 2: the equality rows are A = [I 0], the TotalConstraint
    on the floor) factors the nz - e free rows only, otherwise the (nz + e)-square system.  Per KKT solve: one J^T w and one J v,
    the two triangular sweeps and the closed-form 4 x 4 block inverses (~60 per contact).  `trimmed` (the size-specialised
    instantiations: compile-time nz / neq, pinned): only the columns e .. nz-1 are formed and multiplied - nz rows x (nz - e)
    columns per contact in the formation, nz - e terms in J v.  Nothing of the contact-space pre-factorisation is formed."""
    if trimmed is None:
        trimmed = pinned and (nz, e) in SIZED_SHAPES
    n = (nz - e) if pinned else (nz + e)
    cols = (nz - e) if trimmed else nz
    Fk = 4 * nc * nz * cols + 4 * nc * cols + (2.0 / 3) * n ** 3 + 30 * nc
    prod_t, prod_v = 4 * nc * nz, 4 * nc * cols          # J^T w (dense rows), J v (live columns)
    S = prod_t + prod_v + 2 * n * n + (2 * e * nz if (pinned and not trimmed) else 0) + 60 * nc + 2 * nz
    Rk = prod_t + prod_v + (0 if pinned else 4 * e * nz) + 2 * nz + 30 * nc
    I = Fk + 2 * S + Rk + 60 * nc
    return Fk + S + it * I


def flops_forward_executed_primal(nz, nc, e, it, pinned=True):
    """FLOPs the one-wave-per-scene body-space kernel executes per scene (lcp_primal_step.inc).  Its formation is SPARSE - every
    contact adds its 6 x 6 block (two bodies) to the matrix image: 12 products for the two rows P = B [jc; jt], then 3 per entry -,
    the LU runs over the nz - e free coordinates in the pinned form (nz + e otherwise); per KKT solve: J^T w and J v over the
    contact's six entries, the two sweeps, two closed-form 4 x 4 block inverses (~30 each); per iteration one factorisation, two
    solves, the residuals (one J^T w, one J v, ~30 per contact) and the step lengths (~100 per contact: 20 divisions)."""
    n = (nz - e) if pinned else (nz + e)
    Fk = nc * (12 + 36 * 3 + 30) + (2.0 / 3) * n ** 3
    S = nc * (6 * 3 + 6 * 4 + 60) + 2 * n * n
    Rk = nc * (6 * 3 + 6 * 4 + 30) + 2 * nz
    I = Fk + 2 * S + Rk + 100 * nc
    return Fk + S + it * I


def flops_backward_executed_body_space(nz, nc, e, refine=1, pinned=True):
    """The body-space backward solve (lcp_quad.hip bwd_solve_body): one formation + LU, 1 + `refine` KKT solves (LCP_Q_BWD_REFINE = 1 since round 5; the one-wave-per-scene kernels always took one) and the residual
    products of the refinement steps; the outer products of lcp.py:52-61 are counted with the dense gradient sizes."""
    n = (nz - e) if pinned else (nz + e)
    m = 4 * nc
    Fk = 4 * nc * nz * nz + 4 * nc * nz + (2.0 / 3) * n ** 3 + 30 * nc
    prod = 4 * nc * nz
    S = 2 * prod + 2 * n * n + 60 * nc + 2 * nz
    outer = 2 * nz * nz + 3 * m * nz + m * m + 3 * e * nz
    return Fk + (1 + refine) * S + refine * (2 * prod + 4 * e * nz + 20 * nc) + outer


def flops_backward(nz, m, e):
    k = e + m
    Fk = (2.0 / 3) * m ** 3 + m
    Sk = 4 * nz * nz + 4 * m * nz + 4 * e * nz + 2 * k * k + 6 * m
    return Fk + Sk + 4 * m * nz + m * m + 4 * e * nz + 2 * nz * nz


def bytes_forward(nz, m, e, word=4):
    bytes_in = word * (nz * nz + nz + m * nz + m + e * nz + e + m * m)
    bytes_out = word * (nz + 2 * m + e)
    return bytes_in + bytes_out


def bytes_backward(nz, m, e, word=4):
    rd = word * (nz * nz + m * nz + e * nz + m * m) + word * (nz + 2 * m + e) + word * nz
    wr = word * (nz * nz + nz + m * nz + m + e * nz + e + m * m)
    return rd + wr


def bytes_fused_step(nb, nc, word=4):
    """Fused boundary: contact list + body state in, new velocity + pose out."""
    return word * (14 * nb + 7 * nc) + 8 * nc + word * 6 * nb

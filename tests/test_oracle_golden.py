"""CPU: pin the clean-room oracle against the reference.

* against the committed fixtures (outputs of the unmodified reference recorded by
  oracle/make_golden.py) - runs everywhere;
* live against /root/reference when it is present (marker `reference`).
"""
import pytest
import torch

from oracle import pdipm_oracle as O
from tests import golden_io, parity

STEPS = list(golden_io.all_steps())
IDS = [s[0] for s in STEPS]


@pytest.mark.parametrize("name,st", STEPS, ids=IDS)
def test_forward_matches_reference_fixture(name, st):
    Q, p, G, h, A, b, F = golden_io.lcp_inputs(st)
    sol = O.lcp_forward(Q, p, G, h, A, b, F)
    ex = parity.err_x(sol.x, st["x"], Q, p)
    assert float(ex.max()) < 1e-9, (name, ex)
    # multipliers: unique up to conditioning of the converged KKT system
    assert float(parity.rel_err(sol.z, st["lams"]).max()) < 5e-4
    assert float(parity.rel_err(sol.s, st["slacks"]).max()) < 5e-4
    assert torch.equal(parity.active_sets(sol.z, sol.s), parity.active_sets(st["lams"], st["slacks"]))


def _phys(st):
    ph = {k: st[k][None] for k in parity.PHYS_KEYS}
    ph["c_i1"], ph["c_i2"] = st["c_i1"][None], st["c_i2"][None]
    ph["Je"] = st["Je"].unsqueeze(0) if st["Je"].numel() else None
    return ph


@pytest.mark.parametrize("name,st", STEPS, ids=IDS)
def test_backward_matches_reference_fixture(name, st):
    """See tests/parity.py: dG/dh/dF of the reference are rounding-determined on sticking
    contacts, so pin (1) dp,dQ,dA,db directly, (2) the KKT residual, (3) physical grads."""
    Q, p, G, h, A, b, F = golden_io.lcp_inputs(st)
    sol = O.lcp_forward(Q, p, G, h, A, b, F)
    g = O.lcp_backward(sol, Q, p, G, h, A, b, F, st["cot"])
    grads = {k: g["d" + k] for k in "QpGhAbF"}
    ref = golden_io.ref_grads(st)
    fl = parity.grad_floors(Q, p, st["cot"], st["x"], st["lams"], st.get("nus"))
    errs = parity.err_grads({k: grads[k] for k in "QpAb"}, {k: ref[k] for k in "QpAb"}, fl)
    worst = max(float(e.max()) for e in errs.values())
    assert worst < 1e-6, (name, {k: float(v.max()) for k, v in errs.items()})
    # both are solutions of the system the reference solves, each at its own (z, s)
    for gg, zz, ss in ((grads, sol.z, sol.s), (ref, st["lams"], st["slacks"])):
        res = parity.kkt_backward_residual(Q, G, A, F, zz, ss, st["cot"],
                                           gg["p"], -gg["h"], None if A is None else -gg["b"])
        assert max(float(v.max()) for v in res.values()) < 1e-7, (name, res)
    ph = _phys(st)
    pg = parity.physical_grads(ph, st["dt"], grads, O)
    pg_ref = parity.physical_grads(ph, st["dt"], ref, O)
    sc = parity.free_scales(Q, p, st["cot"])
    floor = parity._n(st["cot"]) * torch.maximum(sc["x_free"], parity._n(st["x"]))
    ep = parity.err_physical(pg, pg_ref, ph, floor)
    assert float(ep.max()) < 1e-6, (name, ep)


@pytest.mark.parametrize("name,st", STEPS, ids=IDS)
def test_assembly_and_new_v_match_reference_fixture(name, st):
    Je = st["Je"].unsqueeze(0) if st["Je"].numel() else None
    new_v, sol, lcp = O.solve_dynamics(
        st["Mdiag"][None], st["v"][None], st["f"][None], st["dt"], st["c_n"][None],
        st["c_p1"][None], st["c_p2"][None], st["c_i1"][None], st["c_i2"][None],
        st["rest"][None], st["fric"][None], Je)
    for mine, ref in zip(lcp, golden_io.lcp_inputs(st)):
        if ref is None:
            assert mine is None
        else:
            assert torch.allclose(mine, ref, rtol=0, atol=1e-12 * max(1.0, float(ref.abs().max())))
    Q, p = lcp[0], lcp[1]
    ev = parity.err_x(-new_v.reshape(1, -1), -st["new_v"].reshape(1, -1), Q, p)
    assert float(ev.max()) < 1e-9


def test_solve_kkt_is_minus_K_inverse():
    """Known-answer: solve_kkt == -K^-1 r for the explicit dense K."""
    g = torch.Generator().manual_seed(3)
    B, nz, m, e = 5, 7, 12, 2
    Lq = torch.randn(B, nz, nz, generator=g, dtype=torch.float64)
    Q = Lq @ Lq.transpose(1, 2) + nz * torch.eye(nz, dtype=torch.float64)
    G = torch.randn(B, m, nz, generator=g, dtype=torch.float64)
    A = torch.randn(B, e, nz, generator=g, dtype=torch.float64)
    F = 0.3 * torch.randn(B, m, m, generator=g, dtype=torch.float64)
    d = torch.rand(B, m, generator=g, dtype=torch.float64) + 0.1
    k = O.pre_factor_kkt(Q, G, F, A)
    O.factor_kkt(k, d)
    rx = torch.randn(B, nz, generator=g, dtype=torch.float64)
    rs = torch.randn(B, m, generator=g, dtype=torch.float64)
    rz = torch.randn(B, m, generator=g, dtype=torch.float64)
    ry = torch.randn(B, e, generator=g, dtype=torch.float64)
    dx, ds, dz, dy = O.solve_kkt(k, d, G, A, rx, rs, rz, ry)
    K = O.dense_kkt_matrix(Q, G, A, F, d)
    r = torch.cat([rx, rs, rz, ry], 1)
    ref = -torch.linalg.solve(K, r.unsqueeze(-1)).squeeze(-1)
    got = torch.cat([dx, ds, dz, dy], 1)
    assert torch.allclose(got, ref, rtol=1e-9, atol=1e-9)


def test_backward_equals_finite_differences_for_symmetric_F():
    """Known-answer: the reference backward formula is the true gradient when F is
    symmetric."""
    g = torch.Generator().manual_seed(5)
    B, nz, m = 3, 5, 6
    Lq = torch.randn(B, nz, nz, generator=g, dtype=torch.float64)
    Q = Lq @ Lq.transpose(1, 2) + nz * torch.eye(nz, dtype=torch.float64)
    G = torch.randn(B, m, nz, generator=g, dtype=torch.float64)
    Fh = torch.randn(B, m, m, generator=g, dtype=torch.float64)
    F = 0.05 * (Fh @ Fh.transpose(1, 2))
    p = torch.randn(B, nz, generator=g, dtype=torch.float64)
    h = torch.rand(B, m, generator=g, dtype=torch.float64)
    cot = torch.randn(B, nz, generator=g, dtype=torch.float64)
    kw = dict(max_iter=30, not_improved_lim=5)
    sol = O.lcp_forward(Q, p, G, h, None, None, F, **kw)
    gr = O.lcp_backward(sol, Q, p, G, h, None, None, F, cot)
    eps = 1e-6
    for j in range(nz):
        dp = torch.zeros_like(p)
        dp[:, j] = eps
        xp = O.lcp_forward(Q, p + dp, G, h, None, None, F, **kw).x
        xm = O.lcp_forward(Q, p - dp, G, h, None, None, F, **kw).x
        fd = ((xp - xm) / (2 * eps) * cot).sum(1)
        assert torch.allclose(fd, gr["dp"][:, j], rtol=1e-4, atol=1e-6)


@pytest.mark.reference
def test_oracle_matches_live_reference_on_random_engine_scenes():
    """Live: run the unmodified reference per scene (batch 1) on synthetic stack scenes
    and compare the vectorised per-scene oracle."""
    from oracle import ref_shim
    mods = ref_shim.load_reference()
    from lcp_physics_amd import scenes
    sc = scenes.make_stack_scenes(B=6, nbox=2, pts_per_interface=2, seed=11, dtype=torch.float64)
    Q, p, G, h, A, b, F = O.assemble_lcp(*sc.assembly_args())
    sol = O.lcp_forward(Q, p, G, h, A, b, F)
    for i in range(Q.shape[0]):
        fn = mods["engines"].LCPFunction(max_iter=10, verbose=-1)
        x = fn(Q[i:i + 1], p[i:i + 1], G[i:i + 1], h[i:i + 1], A[i:i + 1], b[i:i + 1], F[i:i + 1])
        ex = parity.err_x(sol.x[i:i + 1], x, Q[i:i + 1], p[i:i + 1])
        assert float(ex) < 1e-9
        assert torch.equal(parity.active_sets(sol.z[i:i + 1], sol.s[i:i + 1]),
                           parity.active_sets(fn.lams, fn.slacks))

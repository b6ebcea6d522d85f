"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the committed
reference fixtures.  Run with `pytest -m gpu` on an MI355X.

Tolerances (stated, fp32 I/O): err_x <= 1e-4 (scaled, tests/parity.py), gradients <= 1e-4,
contact index sets {i: z_i > s_i} identical wherever the oracle's decision is not a tie.
The fp64-I/O kernel is held to 1e-7 (it follows the reference's native-dtype trajectory).
"""
import numpy as np
import pytest
import torch

from oracle import pdipm_oracle as O
from tests import golden_io, parity

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL_X32 = 1e-4
TOL_G32 = 1e-4


@pytest.fixture(autouse=True, params=["wave64", "generic", "big"])
def kernel_path(request):
    """Every test runs three times: through the library's own choice of kernels (wave-per-scene / four-scenes-per-wave where the
    sizes allow - contact-structured scenes in body space -, the generic kernels otherwise), with the generic kernels forced, and
    with LCP_PATH_CONTACT_SPACE ("big": the contact-space factorisation where the default is a body-space one)."""
    from lcp_physics_amd import _lib
    _lib.set_path(request.param)
    yield request.param
    _lib.set_path("auto")


def _gpu(ts, dtype):
    return [None if t is None else t.to(device=DEV, dtype=dtype).contiguous() for t in ts]


def _solve(lcp, dtype, compute="f64", **kw):
    from lcp_physics_amd.lcp import lcp_solve
    sol = lcp_solve(*_gpu(lcp, dtype), compute=compute, **kw)
    torch.cuda.synchronize()
    return sol


def _decisive(z, s, rel=1e-3, floor=1e-5):
    """Indices where the oracle's z_i > s_i decision is meaningful: not a near-tie, and not a
    degenerate pair where BOTH z_i and s_i have converged to zero (a contact exactly at the
    boundary: which of two 1e-9 numbers is larger is decided by the last PDIPM iteration)."""
    big = torch.maximum(z.abs(), s.abs())
    zs = z.abs().max(dim=1, keepdim=True)[0]
    ss = s.abs().max(dim=1, keepdim=True)[0]
    nondegenerate = torch.maximum(z.abs() / zs, s.abs() / ss) > floor
    return ((z - s).abs() > rel * big) & nondegenerate


def _check_forward(sol, ref, Q, p, tol_x, name="", max_masked=None):
    x = sol.x.double().cpu()
    ex = parity.err_x(x, ref.x, Q.double(), p.double())
    assert float(ex.max()) <= tol_x, (name, "err_x", float(ex.max()), int(ex.argmax()))
    z, s = sol.z.double().cpu(), sol.s.double().cpu()
    dec = _decisive(ref.z, ref.s)
    same = (parity.active_sets(z, s) == parity.active_sets(ref.z, ref.s)) | ~dec
    assert bool(same.all()), (name, "active set", torch.nonzero(~same)[:8].tolist())
    if max_masked is not None:                      # the mask on "bit-exact index sets" is itself gated
        masked = 1.0 - float(dec.float().mean())
        assert masked <= max_masked, (name, "index-set rows masked out as ties", masked)
    return ex


# ------------------------------------------------------------------ golden fixtures (reference)
STEPS = list(golden_io.all_steps())
IDS = [s[0] for s in STEPS]


@pytest.mark.parametrize("name,st", STEPS, ids=IDS)
def test_fp64_kernel_matches_reference_fixture(name, st):
    lcp = golden_io.lcp_inputs(st)
    sol = _solve(lcp, torch.float64)
    Q, p = lcp[0], lcp[1]
    ex = parity.err_x(sol.x.cpu(), st["x"], Q, p)
    assert float(ex.max()) < 1e-7, (name, ex)
    assert float(parity.rel_err(sol.z.cpu(), st["lams"]).max()) < 5e-4
    z, s = sol.z.cpu(), sol.s.cpu()
    dec = _decisive(st["lams"], st["slacks"])
    assert bool(((parity.active_sets(z, s) == parity.active_sets(st["lams"], st["slacks"])) | ~dec).all())
    # LCP_ST_SINGULAR_T (4) is benign: an exact zero pivot once s/z underflows the diagonal of T, the
    # reference's `except: return best` path (pdipm.py:99-102); anything else is a failure.
    assert int(sol.status.cpu().max()) & ~4 == 0


@pytest.mark.parametrize("name,st", STEPS, ids=IDS)
def test_fp32_io_matches_reference_fixture(name, st):
    lcp32 = [None if t is None else t.float() for t in golden_io.lcp_inputs(st)]
    ref = O.lcp_forward(*[None if t is None else t.double() for t in lcp32])
    sol = _solve(lcp32, torch.float32)
    _check_forward(sol, ref, lcp32[0], lcp32[1], TOL_X32, name)
    # and against the reference's own fp64 answer on the un-rounded inputs
    Q, p = golden_io.lcp_inputs(st)[:2]
    ex = parity.err_x(sol.x.double().cpu(), st["x"], Q, p)
    assert float(ex.max()) <= TOL_X32, (name, float(ex.max()))


def _phys(st):
    ph = {k: st[k][None] for k in parity.PHYS_KEYS}
    ph["c_i1"], ph["c_i2"] = st["c_i1"][None], st["c_i2"][None]
    ph["Je"] = st["Je"].unsqueeze(0) if st["Je"].numel() else None
    return ph


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32io"])
@pytest.mark.parametrize("name,st", STEPS, ids=IDS)
def test_backward_matches_reference_fixture(name, st, dtype):
    from lcp_physics_amd.lcp import lcp_backward
    lcp = golden_io.lcp_inputs(st)
    Q, p, G, h, A, b, F = lcp
    sol = _solve(lcp, dtype)
    grads = lcp_backward(sol, st["cot"].to(device=DEV, dtype=dtype))
    torch.cuda.synchronize()
    grads = {k: (None if g is None else g.double().cpu()) for k, g in zip("QpGhAbF", grads)}
    ref = golden_io.ref_grads(st)
    tol = 1e-6 if dtype == torch.float64 else TOL_G32
    fl = parity.grad_floors(Q, p, st["cot"], st["x"], st["lams"], st.get("nus"))
    errs = parity.err_grads({k: grads[k] for k in "QpAb"}, {k: ref[k] for k in "QpAb"}, fl)
    worst = max(float(e.max()) for e in errs.values())
    assert worst < tol, (name, {k: float(v.max()) for k, v in errs.items()})
    res = parity.kkt_backward_residual(Q, G, A, F, sol.z.double().cpu(), sol.s.double().cpu(), st["cot"],
                                       grads["p"], -grads["h"], None if A is None else -grads["b"])
    assert max(float(v.max()) for v in res.values()) < (1e-7 if dtype == torch.float64 else 1e-5), (name, res)
    ph = _phys(st)
    pg = parity.physical_grads(ph, st["dt"], grads, O)
    pg_ref = parity.physical_grads(ph, st["dt"], ref, O)
    sc = parity.free_scales(Q, p, st["cot"])
    floor = parity._n(st["cot"]) * torch.maximum(sc["x_free"], parity._n(st["x"]))
    ep = parity.err_physical(pg, pg_ref, ph, floor)
    assert float(ep.max()) < tol, (name, float(ep.max()))


# ------------------------------------------------------------------ synthetic BASELINE configs
CONFIGS = [("cfg2_stack2x4", 2, 4, 256), ("cfg3_stack4x4", 4, 4, 256), ("faithful_stack4x2", 4, 2, 128)]
# Gates on the two masks of this file, measured with the oracle alone (CPU, fp64) on these seeds:
#   index-set rows that are ties of two converged-to-zero numbers   cfg3 2.2 %   faithful 0 %     cfg2 18.5 %
#   scenes whose backward system the oracle itself solves           cfg3 93.8 %  faithful 95.3 %  cfg2 51.6 %
# (cfg2: four collinear points under ONE light box - the redundant pairs converge to z = s = 0 far more often.)
# A regression of the solver cannot hide behind a growing mask: the shares are asserted.
MAX_MASKED = {"cfg2_stack2x4": 0.20, "cfg3_stack4x4": parity.MAX_MASKED_FRAC + 0.005, "faithful_stack4x2": 0.0}
MIN_WELL_POSED = {"cfg2_stack2x4": 0.5, "cfg3_stack4x4": parity.MIN_WELL_POSED_FRAC, "faithful_stack4x2": parity.MIN_WELL_POSED_FRAC}


@pytest.mark.parametrize("name,nbox,pts,B", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_stack_scenes_forward_parity(name, nbox, pts, B, kernel_path):
    from lcp_physics_amd import scenes
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=1234 + nbox, dtype=torch.float32)
    lcp32 = O.assemble_lcp(*sc.assembly_args())
    ref = O.lcp_forward(*[None if t is None else t.double() for t in lcp32])   # identical inputs, fp64
    sol = _solve(lcp32, torch.float32)
    _check_forward(sol, ref, lcp32[0], lcp32[1], TOL_X32, name, max_masked=MAX_MASKED[name])
    di = (sol.iters.cpu() - ref.iters).abs()
    if kernel_path == "wave64" and name != "cfg3_stack4x4":
        # the library's own choice since round 4: contact-structured scenes are factored in BODY space.  Stacks that converge to
        # rounding inside the ten iterations (two or four points under one or two boxes) then meet the exit tests of pdipm.py:133 on
        # rounding noise of another elimination order: same answers (err_x above), the loop may run an iteration or two longer.
        # Equal counts are asserted where the oracle's formulation runs - kernel_path "big" (LCP_PATH_CONTACT_SPACE) and "generic".
        assert int(di.max()) <= 2, (name, "iteration counts", di.tolist())
    else:
        assert int(di.max()) <= 1 and float((di == 0).float().mean()) >= 0.97, (name, "iteration counts", di.tolist())


@pytest.mark.parametrize("name,nbox,pts,B", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_stack_scenes_backward_parity(name, nbox, pts, B):
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_backward
    B = 64
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=99 + nbox, dtype=torch.float32)
    lcp32 = O.assemble_lcp(*sc.assembly_args())
    lcp64 = [None if t is None else t.double() for t in lcp32]
    ref = O.lcp_forward(*lcp64)
    g = torch.Generator().manual_seed(5)
    cot = torch.randn(B, lcp32[0].shape[1], generator=g, dtype=torch.float32)
    gref = O.lcp_backward(ref, *lcp64, cot.double())
    gref = {k: gref["d" + k] for k in "QpGhAbF"}
    sol = _solve(lcp32, torch.float32)
    grads = lcp_backward(sol, cot.to(DEV))
    torch.cuda.synchronize()
    grads = {k: (None if t is None else t.double().cpu()) for k, t in zip("QpGhAbF", grads)}
    Q, p, G, h, A, b, F = lcp64
    # Scenes that over-converged (mu ~ 1e-17, s/z ~ 1e-19) leave a KKT matrix that is singular to fp64
    # working precision: the reference's own backward solve is junk there (its residual says so), and there
    # is nothing to be on par with.  Compare where the oracle actually solved its system.
    res_o = parity.kkt_backward_residual(Q, G, A, F, ref.z, ref.s, cot.double(), gref["p"], -gref["h"], -gref["b"])
    ok = torch.stack([v for v in res_o.values()]).max(dim=0)[0] < 1e-9
    # ... and where the solution is strictly complementary (a pair with z_i ~ s_i ~ 0 makes the gradient
    # one-sided / the KKT matrix singular: any two correct solvers may then return different answers)
    zs = ref.z.max(dim=1, keepdim=True)[0]
    ss = ref.s.max(dim=1, keepdim=True)[0]
    ok = ok & (torch.maximum(ref.z / zs, ref.s / ss).min(dim=1)[0] > 1e-6)
    assert float(ok.float().mean()) >= MIN_WELL_POSED[name], (name, "too few well-posed scenes", int(ok.sum()))
    fl = parity.grad_floors(Q, p, cot.double(), ref.x, ref.z, ref.y)
    errs = parity.err_grads({k: grads[k] for k in "QpAb"}, {k: gref[k] for k in "QpAb"}, fl)
    worst = max(float(e[ok].max()) for e in errs.values())
    assert worst < TOL_G32, (name, {k: float(v[ok].max()) for k, v in errs.items()})
    res = parity.kkt_backward_residual(Q, G, A, F, sol.z.double().cpu(), sol.s.double().cpu(), cot.double(),
                                       grads["p"], -grads["h"], -grads["b"])
    # (outputs are rounded to fp32 here, which perturbs s/z ~ 1e17 entries: loose bound, the fp64-I/O
    # fixture test holds the same residual to 1e-7)
    assert max(float(v[ok].max()) for v in res.values()) < 2e-2, (name, res)
    ph = {k: v.double() if v.is_floating_point() else v for k, v in sc.phys_dict().items()}
    pg = parity.physical_grads(ph, sc.dt, grads, O)
    pg_ref = parity.physical_grads(ph, sc.dt, gref, O)
    scl = parity.free_scales(Q, p, cot.double())
    floor = parity._n(cot) * torch.maximum(scl["x_free"], parity._n(ref.x))
    # Box stacks have redundant constraint rows (4 collinear normal points; and the tangential rows of two
    # sticking points on one rigid interface are identical), so dlam is non-unique along their null space
    # and only the parameters entering through Q and p have defined gradients (see parity.err_physical).
    # Geometry / friction gradients are pinned on the reference's own scenes in the fixture test above.
    keys = ["Mdiag", "v", "f"]
    ep = parity.err_physical(pg, pg_ref, ph, floor, keys=keys)
    assert float(ep[ok].max()) < TOL_G32, (name, float(ep[ok].max()), int(ep.argmax()))


RANDOM = [("m4", 6, 4, 3), ("m24_noeq", 7, 24, 0), ("m64", 15, 64, 3), ("m96_generic", 20, 96, 2),
          ("m130_ws", 12, 130, 1)]


@pytest.mark.parametrize("name,nz,m,e", RANDOM, ids=[r[0] for r in RANDOM])
def test_random_dense_lcp_forward_backward(name, nz, m, e):
    """Non-physics dense LCPs (non-degenerate): all 7 gradients are well-posed here."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_backward
    B = 24
    lcp32 = scenes.make_random_lcp(B, nz, m, e, seed=nz * 100 + m, dtype=torch.float32)
    lcp64 = [None if t is None else t.double() for t in lcp32]
    kw = dict(max_iter=20, not_improved_lim=3)
    ref = O.lcp_forward(*lcp64, **kw)
    sol = _solve(lcp32, torch.float32, **kw)
    _check_forward(sol, ref, lcp32[0], lcp32[1], TOL_X32, name)
    assert float(parity.rel_err(sol.z.double().cpu(), ref.z).max()) < 1e-4
    g = torch.Generator().manual_seed(1)
    cot = torch.randn(B, nz, generator=g, dtype=torch.float32)
    gref = O.lcp_backward(ref, *lcp64, cot.double())
    grads = lcp_backward(sol, cot.to(DEV))
    torch.cuda.synchronize()
    for k, gt in zip("QpGhAbF", grads):
        r = gref["d" + k]
        if r is None:
            assert gt is None
            continue
        err = parity.rel_err(gt.double().cpu(), r, floor=1e-30)
        assert float(err.max()) < TOL_G32, (name, k, float(err.max()))


def test_fp64_io_random_lcp_tight():
    from lcp_physics_amd import scenes
    lcp64 = scenes.make_random_lcp(16, 9, 32, 3, seed=4, dtype=torch.float64)
    ref = O.lcp_forward(*lcp64)
    sol = _solve(lcp64, torch.float64)
    assert float(parity.rel_err(sol.x.cpu(), ref.x).max()) < 1e-8
    assert float(parity.rel_err(sol.z.cpu(), ref.z).max()) < 1e-7
    assert torch.equal(sol.iters.cpu(), ref.iters)


def test_pure_fp32_compute_mode_is_looser_but_sane():
    """compute='f32' (all-fp32 arithmetic, partial pivoting) is the optional fast mode: it is held
    to the accuracy the reference's own fp32 run reaches (median ~1e-6, tail to 1e-2)."""
    from lcp_physics_amd import scenes
    sc = scenes.make_stack_scenes(B=256, nbox=4, pts_per_interface=4, seed=77, dtype=torch.float32)
    lcp32 = O.assemble_lcp(*sc.assembly_args())
    ref = O.lcp_forward(*[None if t is None else t.double() for t in lcp32])
    sol = _solve(lcp32, torch.float32, compute="f32")
    ex = parity.err_x(sol.x.double().cpu(), ref.x, lcp32[0].double(), lcp32[1].double())
    assert float(ex.median()) < 1e-4 and float(torch.quantile(ex, 0.9)) < 5e-3, (float(ex.median()), float(ex.max()))


# ------------------------------------------------------------------ assembly + fused step
@pytest.mark.parametrize("nbox,pts", [(2, 4), (4, 4), (4, 2)])
def test_assembly_kernel_matches_oracle(nbox, pts):
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics import assemble_contacts
    sc = scenes.make_stack_scenes(B=32, nbox=nbox, pts_per_interface=pts, seed=3, dtype=torch.float32)
    ref = O.assemble_lcp(*sc.assembly_args())
    got = assemble_contacts(sc.to(device=DEV))
    torch.cuda.synchronize()
    for nm, a, r in zip("QpGhAbF", got, ref):
        assert torch.allclose(a.cpu(), r, rtol=1e-6, atol=1e-6 * float(r.abs().max())), nm


def test_assembly_kernel_matches_reference_fixtures():
    from lcp_physics_amd.physics import assemble_contacts
    from lcp_physics_amd.scenes import SceneBatch
    for name, st in STEPS:
        nb = st["v"].shape[0]
        Je = st["Je"][None] if st["Je"].numel() else torch.zeros(1, 0, 3 * nb, dtype=torch.float64)
        sc = SceneBatch(p=st["p"][None], v=st["v"][None], Mdiag=st["Mdiag"][None], f=st["f"][None],
                        rest=st["rest"][None], fric=st["fric"][None], c_n=st["c_n"][None],
                        c_p1=st["c_p1"][None], c_p2=st["c_p2"][None], c_i1=st["c_i1"][None],
                        c_i2=st["c_i2"][None], Je=Je, dt=st["dt"]).to(device=DEV, dtype=torch.float32)
        got = assemble_contacts(sc)
        for nm, a, r in zip("QpGhAbF", got, golden_io.lcp_inputs(st)):
            if r is None:
                assert a is None
            else:
                assert torch.allclose(a.double().cpu(), r, rtol=1e-5, atol=1e-6 * max(1.0, float(r.abs().max()))), (name, nm)


@pytest.mark.parametrize("nbox,pts,B", [(2, 4, 128), (4, 4, 128), (4, 2, 64)])
def test_fused_step_matches_oracle_step(nbox, pts, B):
    """Fused launch (assembly + solve + integrate) vs the oracle solving the SAME LCP: the fp32 LCP data
    the HIP assembly produces (all assembling kernels share one contraction-free builder).  With
    redundant contact points the multipliers are non-unique, so identical inputs are what makes the
    z / s / active-set comparison meaningful."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics import assemble_contacts, fused_step
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=21, dtype=torch.float32)
    scg = sc.to(device=DEV)
    lcp = [None if t is None else t.double().cpu() for t in assemble_contacts(scg)]
    ref = O.lcp_forward(*lcp)
    out = fused_step(scg)
    torch.cuda.synchronize()
    Q, p = lcp[0], lcp[1]
    ex = parity.err_x(-out["v_new"].double().cpu().reshape(B, -1), ref.x, Q, p)
    assert float(ex.max()) <= TOL_X32, float(ex.max())
    p_ref = O.integrate(sc.p.double(), (-ref.x).reshape(B, -1, 3), sc.dt)
    assert torch.allclose(out["p_new"].double().cpu(), p_ref, rtol=1e-5, atol=1e-3)
    # (floor 1e-4: the contact-list path runs the body-space factorisation and may return the best iterate of a converged
    #  solve from one iteration later than the oracle, where a pair on its way to zero is another factor 1e-3 smaller)
    dec = _decisive(ref.z, ref.s, floor=1e-4)
    z, s = out["z"].double().cpu(), out["s"].double().cpu()
    same = (parity.active_sets(z, s) == parity.active_sets(ref.z, ref.s)) | ~dec
    assert bool(same.all()), torch.nonzero(~same)[:8].tolist()
    # and against the independent fp64 assembly of the oracle: velocities only
    new_v, _, lcp_o = O.solve_dynamics(*sc.to(dtype=torch.float64).assembly_args())
    ev = parity.err_x(out["v_new"].double().cpu().reshape(B, -1), new_v.reshape(B, -1), lcp_o[0], lcp_o[1])
    assert float(ev.max()) <= TOL_X32, float(ev.max())


def test_fused_step_equals_assemble_then_solve(kernel_path):
    """The fused kernel and assemble -> dense solve see the same LCP data and run the same arithmetic
    (bit-identical on the wave64 path, where both F operators are exact)."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_solve
    from lcp_physics_amd.physics import assemble_contacts, fused_step
    sc = scenes.make_stack_scenes(B=96, nbox=4, pts_per_interface=4, seed=8, dtype=torch.float32).to(device=DEV)
    lcp = assemble_contacts(sc)
    sol = lcp_solve(*lcp)
    out = fused_step(sc)
    torch.cuda.synchronize()
    ex = parity.err_x(-out["v_new"].double().cpu().reshape(96, -1), sol.x.double().cpu(), lcp[0].double().cpu(), lcp[1].double().cpu())
    assert float(ex.max()) < 1e-6, float(ex.max())
    assert int((out["iters"] - sol.iters).abs().max()) <= 1


def test_mixed_batch_every_scene_served_by_the_right_kernel(kernel_path):
    """One batch holding all three classes: contact-structured + diagonal Q (quad kernel), contact-structured
    with a dense SPD Q (wave64 structured kernel) and unstructured dense LCPs (general kernel).  B = 13 also
    exercises the partially filled last wavefront of the 4-scenes-per-wave kernel."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_backward
    B = 13
    sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=31, dtype=torch.float32)
    Q, p, G, h, A, b, F = [t.clone() for t in O.assemble_lcp(*sc.assembly_args())]
    nz, m = Q.shape[1], G.shape[1]
    g = torch.Generator().manual_seed(3)
    for i in range(4, 8):                                   # dense SPD Q, structure of G / F untouched
        L = torch.randn(nz, nz, generator=g)
        Q[i] = Q[i] + 0.05 * (L @ L.t()) * Q[i].diagonal().min()
    rq, rp, rG, rh, rA, rb, rF = scenes.make_random_lcp(5, nz, m, 3, seed=77, dtype=torch.float32)
    Q[8:], p[8:], G[8:], h[8:], A[8:], b[8:], F[8:] = rq, rp, rG, rh, rA, rb, rF      # unstructured
    lcp32 = [Q, p, G, h, A, b, F]
    lcp64 = [t.double() for t in lcp32]
    ref = O.lcp_forward(*lcp64)
    sol = _solve(lcp32, torch.float32)
    _check_forward(sol, ref, Q, p, TOL_X32, "mixed")
    if kernel_path == "wave64":
        from lcp_physics_amd import _lib
        import ctypes
        # classification flags live in the workspace: meta[0] of each scene (see lcp_wave_common.h `Ws`)
        stride = (_lib.workspace_bytes(B, nz, m, 3, _lib.COMPUTE_F64) - ((B * 4 + 255) & ~255) - 256) // B     # (the tail holds lcp_big's classes and the layout tag)
        flags = sol.ws[:B * stride].view(torch.float64).reshape(B, stride // 8)[:, 5080].cpu()
        assert flags.tolist() == [2.0] * 4 + [1.0] * 4 + [0.0] * 5
    cot = torch.randn(B, nz, generator=g, dtype=torch.float32)
    gref = O.lcp_backward(ref, *lcp64, cot.double())
    grads = lcp_backward(sol, cot.to(DEV))
    torch.cuda.synchronize()
    fl = parity.grad_floors(lcp64[0], lcp64[1], cot.double(), ref.x, ref.z, ref.y)
    errs = parity.err_grads({"p": grads[1].double().cpu(), "Q": grads[0].double().cpu()}, {"p": gref["dp"], "Q": gref["dQ"]}, fl)
    assert max(float(e.max()) for e in errs.values()) < TOL_G32, {k: float(v.max()) for k, v in errs.items()}


@pytest.mark.parametrize("B", [1, 3, 6])
def test_fused_step_partial_wavefronts(B, kernel_path):
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics import assemble_contacts, fused_step
    sc = scenes.make_stack_scenes(B=B, nbox=2, pts_per_interface=2, seed=40 + B, dtype=torch.float32).to(device=DEV)
    lcp = [None if t is None else t.double().cpu() for t in assemble_contacts(sc)]
    ref = O.lcp_forward(*lcp)
    out = fused_step(sc)
    torch.cuda.synchronize()
    ex = parity.err_x(-out["v_new"].double().cpu().reshape(B, -1), ref.x, lcp[0], lcp[1])
    assert float(ex.max()) <= TOL_X32
    # these small stacks converge to rounding (resid ~ 1e-12) before the iteration limit: the exit tests of pdipm.py:133 then
    # compare numbers that are rounding noise, and the body-space factorisation of the contact-list path (another elimination
    # order than the oracle's) can meet them an iteration or two apart.  Exact equality is asserted where the oracle's
    # formulation runs: the dense boundary (test_stack_scenes_forward_parity) and the forced contact-space path below.
    assert int((out["iters"].cpu() - ref.iters).abs().max()) <= 2
    from lcp_physics_amd import _lib
    _lib.set_path("big")
    try:
        out_cs = fused_step(sc)
        torch.cuda.synchronize()
    finally:
        _lib.set_path(kernel_path)
    if kernel_path != "generic":
        assert torch.equal(out_cs["iters"].cpu(), ref.iters)


# ------------------------------------------------------------------ full BASELINE size: properties
def test_full_size_config3_properties():
    """B = 4096 x 16 contacts (nineq 64): size-independent properties + sampled oracle parity."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics import assemble_contacts
    B = 4096
    sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=1236, dtype=torch.float32)
    lcp = assemble_contacts(sc.to(device=DEV))
    from lcp_physics_amd.lcp import lcp_solve
    sol = lcp_solve(*lcp)
    sol2 = lcp_solve(*lcp)
    torch.cuda.synchronize()
    assert torch.equal(sol.x, sol2.x) and torch.equal(sol.z, sol2.z)          # deterministic / idempotent
    x, z, s = sol.x.double(), sol.z.double(), sol.s.double()
    Q, p, G, h, A, b, F = [t.double() for t in lcp]
    assert bool((z > 0).all()) and bool((s > 0).all())                        # interior iterates
    mv = lambda M, v: torch.bmm(M, v.unsqueeze(-1)).squeeze(-1)
    rz = mv(G, x) + s - h - mv(F, z)
    scale = torch.linalg.solve(Q, p.unsqueeze(-1)).squeeze(-1).norm(dim=1)
    assert float((rz.norm(dim=1) / (G.norm(dim=(1, 2)) * scale)).max()) < 1e-4   # primal feasibility
    assert float((mv(A, x).norm(dim=1) / scale).max()) < 1e-5                  # pinned floor
    comp = (s * z).sum(1) / (64 * (z.norm(dim=1) * scale + 1e-30))
    assert float(comp.median()) < 1e-6
    assert int(sol.status.max()) & ~4 == 0
    idx = torch.arange(0, B, 37)
    sub = [t[idx].cpu() for t in (Q, p, G, h, A, b, F)]
    ref = O.lcp_forward(*sub)
    ex = parity.err_x(sol.x[idx].double().cpu(), ref.x, sub[0], sub[1])
    assert float(ex.max()) <= TOL_X32


def test_pile_config5_generic_path_runs_small_batch():
    """Config 5 shape (nineq 256, nz 33): exercises the workspace-resident T path."""
    from lcp_physics_amd import scenes
    sc = scenes.make_pile_scenes(B=8, seed=5, dtype=torch.float32)
    lcp32 = O.assemble_lcp(*sc.assembly_args())
    ref = O.lcp_forward(*[None if t is None else t.double() for t in lcp32])
    sol = _solve(lcp32, torch.float32)
    _check_forward(sol, ref, lcp32[0], lcp32[1], TOL_X32, "pile")


# ------------------------------------------------------------------ the autograd op and the engine
def test_lcpfunction_autograd_surface():
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import LCPFunction
    lcp = scenes.make_random_lcp(4, 6, 12, 2, seed=9, dtype=torch.float64)
    ins = [t.clone().requires_grad_(True) for t in lcp]            # CPU float64 like the reference
    fn = LCPFunction(max_iter=20)
    x = fn(*ins)
    assert x.shape == (4, 6) and x.dtype == torch.float64 and not x.is_cuda
    assert fn.lams.shape == (4, 12) and fn.slacks.shape == (4, 12) and fn.nus.shape == (4, 2)
    assert (fn.neq, fn.nineq, fn.nz) == (2, 12, 6)
    g = torch.Generator().manual_seed(2)
    cot = torch.randn(4, 6, generator=g, dtype=torch.float64)
    x.backward(cot)
    ref = O.lcp_forward(*lcp, max_iter=20)
    gref = O.lcp_backward(ref, *lcp, cot)
    assert float(parity.rel_err(x.detach(), ref.x).max()) < 1e-8
    for k, t in zip("QpGhAbF", ins):
        assert float(parity.rel_err(t.grad, gref["d" + k]).max()) < 1e-6, k
    # no-equality form: A = b = torch.tensor([])  (engines.py:59-60)
    x2 = LCPFunction()(lcp[0], lcp[1], lcp[2], lcp[3], torch.tensor([]), torch.tensor([]), lcp[6])
    ref2 = O.lcp_forward(lcp[0], lcp[1], lcp[2], lcp[3], None, None, lcp[6])
    assert float(parity.rel_err(x2, ref2.x).max()) < 1e-8


def test_singular_Q_raises_like_reference():
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import LCPFunction
    Q, p, G, h, A, b, F = scenes.make_random_lcp(2, 5, 8, 0, seed=1, dtype=torch.float64)
    Q = torch.zeros_like(Q)
    with pytest.raises(RuntimeError, match="Cannot perform LU factorization on Q"):
        LCPFunction()(Q, p, G, h, torch.tensor([]), torch.tensor([]), F)


@pytest.mark.parametrize("name,st", STEPS[::4], ids=IDS[::4])
def test_engine_plugin_reproduces_reference_new_v(name, st):
    """`HipPdipmEngine` driven by the recorded answers of the reference's real `World` (tests/world_io.py::RecordedWorld)
    returns the `new_v` the reference's own engine returned at that step (fp32 contact data on the device path)."""
    from lcp_physics_amd.physics import HipPdipmEngine
    from tests.world_io import RecordedWorld
    eng = HipPdipmEngine()                      # zero-arg construction, as world.py:26 does
    new_v = eng.solve_dynamics(RecordedWorld(st), st["dt"])
    assert new_v.dtype == torch.float64 and not new_v.is_cuda       # comes back where the world keeps its state
    lcp = golden_io.lcp_inputs(st)
    ev = parity.err_x(-new_v.reshape(1, -1), -st["new_v"].reshape(1, -1), lcp[0], lcp[1])
    assert float(ev.max()) < 1e-4, (name, float(ev.max()))
    assert int(eng.last["status"].item()) & ~4 == 0


def test_engine_plugin_follows_every_recorded_step_of_the_ball_and_floor_world():
    """BASELINE configs[0] end to end: every LCP step the reference's `World` recorded for the falling ball + floor scene
    (the ball landing, bouncing and coming to rest), through both plug-ins."""
    from lcp_physics_amd.physics import HipFusedEngine, HipPdipmEngine
    from tests.world_io import RecordedWorld
    steps = golden_io.load_steps("ball_floor")
    assert len(steps) >= 4
    for Eng in (HipPdipmEngine, HipFusedEngine):
        eng = Eng()
        for k, st in enumerate(steps):
            new_v = eng.solve_dynamics(RecordedWorld(st), st["dt"])
            lcp = golden_io.lcp_inputs(st)
            ev = parity.err_x(-new_v.reshape(1, -1), -st["new_v"].reshape(1, -1), lcp[0], lcp[1])
            assert float(ev.max()) < 1e-4, (Eng.__name__, k, float(ev.max()))


@pytest.mark.parametrize("name,st", STEPS[::5], ids=IDS[::5])
def test_engine_plugin_is_differentiable_like_the_reference(name, st, kernel_path):
    """`loss.backward()` through `HipPdipmEngine.solve_dynamics` reaches the world's leaves (masses, velocities, forces,
    restitution, friction, contact frame) with the gradients the reference's autograd produces: its recorded
    `LCPFunction.backward` outputs (lcp.py:37-64) contracted through the engine assembly (engines.py:31-32,50-74)."""
    from lcp_physics_amd.physics import HipPdipmEngine
    from tests.world_io import RecordedWorld
    if kernel_path == "generic":
        pytest.skip("the analytic step backward belongs to the quad / big kernel families (lcp_step_backward_f32)")
    world = RecordedWorld(st, leaf=True)
    new_v = HipPdipmEngine().solve_dynamics(world, st["dt"])
    nz = new_v.numel()
    cot_x = st["cot"].reshape(-1)[:nz].double()                 # the fixture's cotangent is on x = -new_v
    (new_v * (-cot_x)).sum().backward()
    lv = world.leaves()
    assert all(t.grad is not None for t in lv.values())
    dense = {k: g for k, g in golden_io.ref_grads(st).items() if g is not None}
    ph = {k: v.detach().unsqueeze(0) for k, v in lv.items()}
    ph["c_i1"], ph["c_i2"], ph["Je"] = st["c_i1"].unsqueeze(0), st["c_i2"].unsqueeze(0), st["Je"].unsqueeze(0).double()
    ref = parity.physical_grads(ph, st["dt"], dense, O)
    # the same joint metric as test_backward_matches_reference_fixture (err_physical: blocks weighted by the parameter
    # norms, floored by |cot| x free motion), every physical key, fp32 tolerance
    lcp = golden_io.lcp_inputs(st)
    sc = parity.free_scales(lcp[0], lcp[1], st["cot"])
    floor = parity._n(st["cot"]) * torch.maximum(sc["x_free"], parity._n(st["x"]))
    pg = {k: lv[k].grad.unsqueeze(0) for k in parity.PHYS_KEYS}
    ep = parity.err_physical(pg, ref, ph, floor)
    assert float(ep.max()) < TOL_G32, (name, float(ep.max()))
    # the joint Jacobian: d(loss)/d(world.Je()) IS the reference's recorded dA (lcp.py:57)
    # (dA = dnu x^T + nu dx^T: on a resting stack x and dx vanish on the pinned body and dA is 1e-12 in fp64 - the fp32 solve
    #  leaves |nu| x its own dx error there, hence the absolute term)
    dA = dense["A"].reshape(world._Je.shape).double()
    eA = float((world._Je.grad - dA).abs().max())
    assert world._Je.grad is not None and eA <= 2e-4 * float(dA.abs().max()) + 1e-6 * float(st["nus"].abs().max()) * float(floor.max()), (name, eA)


@pytest.mark.parametrize("name,st", STEPS[::5], ids=IDS[::5])
def test_engine_plugin_post_stabilization_is_differentiable_like_the_reference(name, st):
    """`HipPdipmEngine.post_stabilization(world)` as a node of the autograd graph (what a reference `World(post_stab=True)` with the
    engine swapped differentiates, experiments/inference.py): gradients at the world's leaves against the fp64 oracle's solve of
    engines.py:80-116, `lcp.py:37-64` and autograd through the assembly."""
    from lcp_physics_amd import _lib
    from lcp_physics_amd.physics import HipPdipmEngine
    from tests.world_io import RecordedWorld
    _lib.set_path("auto")                   # (the backward reads the iterate the body-space kernel keeps: the default routing)
    world = RecordedWorld(st, leaf=True)
    dp = HipPdipmEngine().post_stabilization(world)
    cot = torch.randn(dp.shape, generator=torch.Generator().manual_seed(11), dtype=dp.dtype)
    (dp * cot).sum().backward()
    nb = st["v"].shape[0]
    leaf = lambda t: t.detach().double().clone().unsqueeze(0).requires_grad_(True)
    Md, v, rest, Je = leaf(world._Md), leaf(world._v), leaf(world._rest), leaf(world._Je)
    cn, cp1, cp2 = leaf(world._cn), leaf(world._cp1), leaf(world._cp2)
    lcp = O.assemble_post_stabilization(Md, v, cn, cp1, cp2, st["c_i1"].unsqueeze(0), st["c_i2"].unsqueeze(0), rest, Je)
    det = [None if t is None else t.detach() for t in lcp]
    sol = O.lcp_forward(*det)
    gr = O.lcp_backward(sol, *det, -cot.reshape(1, -1).double())               # dp = -x
    outs, cots = [], []
    for t, key in zip(lcp, ("dQ", "dp", "dG", "dh", "dA", "db", "dF")):
        if t is not None and t.requires_grad and gr[key] is not None:
            outs.append(t); cots.append(gr[key])
    torch.autograd.backward(outs, cots)
    # ten iterations do not always converge on this LCP (a resting contact has rhs ~ 0, tests/test_world_oracle.py): the gradients
    # are compared where the oracle's iterate is a solution with strictly complementary rows
    zs, ss = sol.z.max(dim=1, keepdim=True)[0], sol.s.max(dim=1, keepdim=True)[0]
    converged = float((sol.s * sol.z).abs().max()) <= 1e-9 * max(1.0, float((zs * ss).max()))
    strict = float(torch.maximum(sol.z / zs, sol.s / ss).min()) > 1e-6
    for t in (world._Md, world._v, world._rest, world._Je, world._cn, world._cp1, world._cp2):
        assert t.grad is not None and bool(torch.isfinite(t.grad).all())
    if not (converged and strict):
        pytest.skip("the oracle's ten iterations did not reach a strictly complementary solution of this step")
    assert float((dp.detach().double() + sol.x[0]).abs().max()) <= 1e-4 * max(1.0, float(sol.x.abs().max()))
    vscale = max(float(v.grad.abs().max()), 1e-30)
    for key, got, ref in (("Mdiag", world._Md.grad, Md.grad), ("v", world._v.grad, v.grad), ("rest", world._rest.grad, rest.grad),
                          ("Je", world._Je.grad, Je.grad), ("c_n", world._cn.grad, cn.grad), ("c_p1", world._cp1.grad, cp1.grad),
                          ("c_p2", world._cp2.grad, cp2.grad)):
        assert got is not None, key
        r = ref[0] if ref is not None else torch.zeros_like(got)
        scale = max(float(r.abs().max()), 1e-6 * vscale)
        bound = 2e-3 if key in ("c_n", "c_p1", "c_p2") else 2e-4
        assert float((got.double().reshape(r.shape) - r).abs().max()) <= bound * scale, (name, key, float((got.double().reshape(r.shape) - r).abs().max()), scale)


@pytest.mark.parametrize("name,st", STEPS[::5], ids=IDS[::5])
def test_fused_engine_plugin_runs_both_branches_and_post_stabilization_on_the_device(name, st):
    """`HipFusedEngine` (the non-differentiable plug-in for the reference's `World`): contact branch against the
    reference's recorded new_v, the no-contact branch (`engines.py:36-50`) and `post_stabilization`
    (`engines.py:80-116`) against the oracle - all three through the device entry points, batch of one."""
    from oracle import world_oracle as W
    from lcp_physics_amd.physics import HipFusedEngine
    from tests.world_io import RecordedWorld
    eng = HipFusedEngine()
    new_v = eng.solve_dynamics(RecordedWorld(st), st["dt"])
    lcp = golden_io.lcp_inputs(st)
    ev = parity.err_x(-new_v.reshape(1, -1), -st["new_v"].reshape(1, -1), lcp[0], lcp[1])
    assert float(ev.max()) < 1e-4, (name, "contact branch", float(ev.max()))       # (fp32 contact data on this path)
    n = lambda k: st[k].double().numpy()
    Md = n("Mdiag")
    free = eng.solve_dynamics(RecordedWorld(st, with_contacts=False), st["dt"]).reshape(-1, 3).double().numpy()
    ref = W.solve_dynamics(Md, n("v"), n("f"), float(st["dt"]), [], n("rest"), n("fric"), n("Je"))
    assert np.abs(free - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (name, "no-contact branch")
    world = RecordedWorld(st)
    dp = eng.post_stabilization(world).reshape(-1, 3).double().numpy()
    cs = [((c[0][0].detach().numpy(), c[0][1].detach().numpy(), c[0][2].detach().numpy(), float(c[0][3])), c[1], c[2])
          for c in world.contacts]
    ref = W.post_stabilization(Md, n("v"), cs, n("rest"), n("Je"))
    assert np.abs(dp - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), (name, "post_stabilization", np.abs(dp - ref).max())


# ------------------------------------------------------------------ the operator boundary at every size and dtype
def test_fp64_io_takes_the_fast_kernels_and_agrees_with_the_generic_ones(kernel_path):
    """`lcp_pdipm_forward_f64 / _backward_f64` (the reference's native dtype, physics/utils.py:34) run the wave / quad kernels
    with fp64 loads and stores; the same inputs through the forced generic kernels give the same answer (two different
    kernel families, both within 1e-7 of the oracle)."""
    from lcp_physics_amd import _lib, scenes
    from lcp_physics_amd.lcp import lcp_backward
    sc = scenes.make_stack_scenes(B=64, nbox=4, pts_per_interface=4, seed=4242, dtype=torch.float64)
    lcp64 = [None if t is None else t.double() for t in O.assemble_lcp(*sc.assembly_args())]
    ref = O.lcp_forward(*lcp64)
    sol = _solve(lcp64, torch.float64)
    ex = parity.err_x(sol.x.cpu(), ref.x, lcp64[0], lcp64[1])
    assert float(ex.max()) < 1e-7, (kernel_path, float(ex.max()))
    dec = parity.decisive_rows(ref.z, ref.s)
    assert bool(((parity.active_sets(sol.z.cpu(), sol.s.cpu()) == parity.active_sets(ref.z, ref.s)) | ~dec).all())
    cot = torch.randn(64, lcp64[0].shape[1], generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    grads = lcp_backward(sol, cot.to(DEV))
    gref = O.lcp_backward(ref, *lcp64, cot)
    ok = parity.backward_well_posed(lcp64[0], lcp64[2], lcp64[4], lcp64[6], ref, cot, gref)
    fl = parity.grad_floors(lcp64[0], lcp64[1], cot, ref.x, ref.z, ref.y)
    # These fp64 solves converge to rounding inside the ten iterations (residuals of 1e-12 .. 1e-13): pdipm.py:107-132 then compares
    # residuals that are rounding noise, and which of the last iterates is kept as "best" is decided by that noise - in the oracle as in
    # the kernel.  With four collinear points per interface successive iterates differ by per cent in the multipliers (a null space;
    # x agrees to 1e-14 either way), and the backward system is built from them (its own residual is no gate here: s_i / z_i of a
    # converged row is a ratio of two 1e-17 numbers - tests/parity.py::headline_report).  The gradients are compared where kernel and
    # oracle kept the same iterate.  (Until round 4 that was all 64 scenes because the contact-space kernel happened to round like the oracle, operation for
    # operation - nothing a compiler version or a code layout owes anybody: tools/experiments/best_iterate_ties.py.)
    g64 = {k: (None if g is None else g.double().cpu()) for k, g in zip("QpGhAbF", grads)}
    zk, sk = sol.z.double().cpu(), sol.s.double().cpu()
    same = (parity._n(zk - ref.z) / parity._n(ref.z)) < 2e-3          # (the same iterate up to its null-space drift; another iterate: per cent)
    print("well-posed scenes where kernel and oracle kept the same iterate:", int((ok & same).sum()), "of", int(ok.sum()))
    assert int((ok & same).sum()) >= 40
    errs = parity.err_grads({k: g64[k] for k in "QpAb"}, {k: gref["d" + k] for k in "QpAb"}, fl)
    # ... and a gate that does not select on the outcome: on EVERY scene all seven gradients finite, and the kernel's backward
    # against the ORACLE'S BACKWARD EVALUATED AT THE KERNEL'S OWN ITERATE - the same linear system (lcp.py:44-50), two solvers - on every
    # scene where that system determines its solution: well-posed by the oracle's own residual AND stable under a relative perturbation
    # of 1e-7 of (z, s) (a converged contact pair with z_i ~ s_i ~ 1e-17 leaves a matrix singular to working precision: two solvers that
    # both satisfy it to 1e-16 differ by per cent - tools/experiments/fp64_io_backward_diag.py).  The conditioning test uses the oracle alone.
    assert all(bool(torch.isfinite(g).all()) for g in g64.values() if g is not None)
    import copy
    def oracle_backward_at(x, y, z, s_):
        at = copy.copy(ref)
        at.x, at.y, at.z, at.s = x, y, z, s_
        g = O.lcp_backward(at, *lcp64, cot)
        return g, parity.backward_well_posed(lcp64[0], lcp64[2], lcp64[4], lcp64[6], at, cot, g)
    xk, yk = sol.x.double().cpu(), (None if ref.y is None else sol.y.double().cpu())
    gat, ok_at = oracle_backward_at(xk, yk, zk, sk)
    gen = torch.Generator().manual_seed(11)
    wob = lambda t: t * (1 + 1e-7 * torch.randn(t.shape, generator=gen, dtype=torch.float64))
    gat2, ok_at2 = oracle_backward_at(xk, yk, wob(zk), wob(sk))
    moved = parity.err_grads({k: gat2["d" + k] for k in "QpAb"}, {k: gat["d" + k] for k in "QpAb"}, fl)
    stable = ok_at & ok_at2 & (torch.stack(list(moved.values())).max(dim=0)[0] < 1e-5)
    errs_at = parity.err_grads({k: g64[k] for k in "QpAb"}, {k: gat["d" + k] for k in "QpAb"}, fl)
    print("oracle backward at the kernel's own iterate: determined on", int(stable.sum()), "of 64 scenes; errors there",
          {k: float(v[stable].max()) for k, v in errs_at.items()})
    assert int(stable.sum()) >= 56, int(stable.sum())
    # (3e-10 on the fast kernels since their backward repeats a factorisation that met a noise pivot - factor_bwd_q, round 5; 7e-2 on one scene before)
    assert max(float(v[stable].max()) for v in errs_at.values()) < 1e-7, {k: float(v[stable].max()) for k, v in errs_at.items()}
    ok = ok & same
    assert max(float(e[ok].max()) for e in errs.values()) < 1e-6, {k: float(v[ok].max()) for k, v in errs.items()}
    if kernel_path != "generic":                      # the fast path really is a different kernel: its timing says so
        big = scenes.make_stack_scenes(B=2048, nbox=4, pts_per_interface=4, seed=1, dtype=torch.float64)
        lcpb = _gpu([None if t is None else t.double() for t in O.assemble_lcp(*big.assembly_args())], torch.float64)
        from lcp_physics_amd.lcp import lcp_solve
        def timed():
            s0 = lcp_solve(*lcpb)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lcp_solve(*lcpb, ws=s0.ws); e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1)
        fast = timed()
        _lib.set_path("generic")
        try:
            slow = timed()
        finally:
            _lib.set_path(kernel_path)
        assert fast < 0.25 * slow, (fast, slow)


@pytest.mark.parametrize("nbox,pts,dtype,dense_q", [(3, 2, torch.float64, False), (4, 4, torch.float64, False), (2, 2, torch.float64, False),
                                                     (4, 2, torch.float32, False), (4, 4, torch.float64, True), (2, 4, torch.float64, True)])
def test_backward_on_converged_solves_agrees_with_the_oracle_at_the_kernels_own_iterate(kernel_path, nbox, pts, dtype, dense_q):
    """Small stacks converge to rounding inside the ten iterations: `s / z` of the active rows is 1e-13 .. 1e-17 and the backward matrix
    of lcp.py:44-46 is singular to working precision wherever contact points are redundant.  Whatever iterate the kernel kept, its
    backward must agree with the oracle's backward evaluated AT THAT ITERATE (`parity.own_iterate_backward`) on every scene where
    the oracle alone says the system determines its solution - 1024 scenes per shape, every kernel family that serves the size.
    `dense_q`: a non-diagonal SPD Q - contact structure without the diagonal Q the four-scenes-per-wave kernels want: lcp_*_wave_any.
    Before round 5 the contact-space kernels (one scene of 64 at 4 x 4 points, fp64 tensors: dx off by 7 %), the generic kernels
    (one of 1024 at 3 x 2 points: off by 2.7) and the wave-per-scene kernels (dense Q, 4 x 4 points: multipliers of 1e32) divided
    by a pivot that was rounding noise (profiles/r05_own_iterate_probe.txt); each now repeats such a factorisation with s / z floored."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_backward
    B = 1024
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=4242, dtype=dtype)
    lcp = list(O.assemble_lcp(*sc.assembly_args()))
    if dense_q:
        Q = lcp[0]
        E = torch.randn(Q.shape, generator=torch.Generator().manual_seed(5), dtype=Q.dtype) * 0.02
        dq = torch.diagonal(Q, dim1=1, dim2=2).sqrt()
        lcp[0] = Q + (E + E.transpose(1, 2)) * dq.unsqueeze(2) * dq.unsqueeze(1)
    lcp64 = [None if t is None else t.double() for t in lcp]
    ref = O.lcp_forward(*lcp64)
    sol = _solve(lcp, dtype)
    cot = torch.randn(B, lcp64[0].shape[1], generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    grads = lcp_backward(sol, cot.to(device=DEV, dtype=dtype))
    g64 = {k: (None if g is None else g.double().cpu()) for k, g in zip("QpGhAbF", grads)}
    assert all(bool(torch.isfinite(g).all()) for g in g64.values() if g is not None)
    fl = parity.grad_floors(lcp64[0], lcp64[1], cot, ref.x, ref.z, ref.y)
    rep = parity.own_iterate_backward(O, lcp64, ref, cot, sol.x.double().cpu(), sol.z.double().cpu(), sol.s.double().cpu(), g64, fl)
    print(kernel_path, nbox, pts, dtype, dense_q, rep)
    assert rep["bwd_own_iterate_determined_scenes"] >= 0.5 * B, rep      # (642 .. 1010 of 1024: the body-space loop runs a step further into convergence on fp32 data)
    assert rep["bwd_own_iterate_err_max"] <= (1e-7 if dtype == torch.float64 else 1e-5), rep


def _pile_lcp(B, seed):
    from lcp_physics_amd import scenes
    sc = scenes.make_pile_scenes(B=B, seed=seed, dtype=torch.float32)
    return sc, O.assemble_lcp(*sc.assembly_args())


def test_dense_boundary_reaches_the_big_kernel_at_config5_sizes(kernel_path):
    """`LCPFunction`-level inputs of BASELINE config 5 (nz 33, nineq 256, neq 3): the dense entry points classify every scene on
    the device and serve the contact-structured ones from lcp_big.hip (blocked MFMA LU), the rest - here two scenes whose F was
    perturbed - from the generic kernels, in ONE call.  Forward: err_x <= 1e-4 and index sets against the oracle; backward: the
    well-defined gradients against the oracle and all seven against the generic kernels' (same multipliers, fp32 stores)."""
    from lcp_physics_amd import _lib
    from lcp_physics_amd.lcp import lcp_backward
    B = 10
    sc, lcp32 = _pile_lcp(B, 77)
    lcp32 = [None if t is None else t.clone() for t in lcp32]
    nc = sc.nc
    lcp32[6][3, 5, 7] = 0.25                          # scenes 3 and 8 lose the contact structure: they are general LCPs now
    lcp32[6][8, 3 * nc + 2, 2 * nc] = -0.5
    lcp64 = [None if t is None else t.double() for t in lcp32]
    ref = O.lcp_forward(*lcp64)
    sol = _solve(lcp32, torch.float32)
    _check_forward(sol, ref, lcp32[0], lcp32[1], TOL_X32, "config5 dense")
    assert int(sol.status.cpu().max()) & ~4 == 0
    cot = torch.randn(B, lcp32[0].shape[1], generator=torch.Generator().manual_seed(11), dtype=torch.float32)
    grads = [None if g is None else g.double().cpu() for g in lcp_backward(sol, cot.to(DEV))]
    torch.cuda.synchronize()
    if kernel_path != "generic":
        _lib.set_path("generic")
        try:
            solg = _solve(lcp32, torch.float32)
            gg = [None if g is None else g.double().cpu() for g in lcp_backward(solg, cot.to(DEV))]
            torch.cuda.synchronize()
        finally:
            _lib.set_path(kernel_path)
        assert float((sol.x.cpu() - solg.x.cpu()).abs().max()) <= 1e-5 * max(1.0, float(solg.x.abs().max()))
        for k, a, b in zip("QpGhAbF", grads, gg):
            if k in "QpAb":                           # (dG, dh, dF of redundant piles are rounding-determined: tests/parity.py)
                scale = float(b.abs().max())
                assert float((a - b).abs().max()) <= 1e-4 * max(scale, 1e-12), (k, float((a - b).abs().max()), scale)
    gref = O.lcp_backward(ref, *lcp64, cot.double())
    ok = parity.backward_well_posed(lcp64[0], lcp64[2], lcp64[4], lcp64[6], ref, cot.double(), gref)
    fl = parity.grad_floors(lcp64[0], lcp64[1], cot.double(), ref.x, ref.z, ref.y)
    errs = parity.err_grads({k: g for k, g in zip("QpGhAbF", grads) if k in "QpAb"}, {k: gref["d" + k] for k in "QpAb"}, fl)
    assert int(ok.sum()) >= B // 2, ("too few scenes whose backward system the oracle itself solves", int(ok.sum()), B)
    assert max(float(e[ok].max()) for e in errs.values()) < TOL_G32, {k: float(v[ok].max()) for k, v in errs.items()}
    # the rank-1 structure of the dense gradients (lcp.py:53-56): dF = -dlam (x) lam, dh = -dlam
    dF, dh = grads[6], grads[3]
    lam = sol.z.double().cpu()
    rebuilt = dh.unsqueeze(2) * lam.unsqueeze(1)
    assert float((dF - rebuilt).abs().max()) <= 1e-5 * max(1.0, float(dF.abs().max()))


def test_dense_boundary_speed_at_config5_sizes(kernel_path):
    """The point of the routing: LCPFunction-level config 5 no longer runs at the generic kernels' 2.3 k sim steps/s."""
    from lcp_physics_amd.lcp import lcp_solve
    if kernel_path == "generic":
        pytest.skip("forced generic kernels")
    B = 1024
    _, lcp32 = _pile_lcp(B, 5)
    g = _gpu(lcp32, torch.float32)
    s0 = lcp_solve(*g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lcp_solve(*g, ws=s0.ws); e1.record(); torch.cuda.synchronize()
    rate = B / (e0.elapsed_time(e1) * 1e-3)
    print("dense config-5 forward: %.0f sim steps/s" % rate)
    assert rate > 1e5, rate


def test_opt_in_adjoint_backward_matches_finite_differences_where_the_references_formula_does_not():
    """`LCP_BWD_ADJOINT` / `LCPFunction(adjoint_backward=True)` (VERDICT r05 missing 6): the backward solves with K^T.  On a
    contact LCP - F is not symmetric (engines.py:69-73) - and converged solves (max_iter 30) the gradient d(c . x)/dp of the adjoint
    form agrees with central finite differences of the forward and with the oracle's `adjoint=True`; the reference's own formula
    (lcp.py:46-50: K, the default and the parity target) is measurably off on the same scenes - that is its documented defect, reproduced."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_backward, lcp_solve
    from lcp_physics_amd.physics import assemble_contacts
    B = 16
    sc = scenes.make_stack_scenes(B=B, nbox=2, pts_per_interface=1, seed=77, dtype=torch.float64)     # one point per interface: unique multipliers
    lcp = [None if t is None else t.double() for t in assemble_contacts(sc.to(device=DEV, dtype=torch.float32))]
    Q, p, G, h, A, b, F = lcp
    assert float((F - F.transpose(1, 2)).abs().max()) > 0.1                                           # F is not symmetric
    cot = torch.randn(B, Q.shape[1], generator=torch.Generator().manual_seed(5), dtype=torch.float64).to(DEV)
    kw = dict(max_iter=30, path="generic")
    sol = lcp_solve(Q, p, G, h, A, b, F, **kw)
    g_adj = lcp_backward(sol, cot, adjoint=True)[1].cpu()
    g_ref = lcp_backward(sol, cot)[1].cpu()
    # central differences of L = cot . x(p)
    fd = torch.zeros(B, Q.shape[1], dtype=torch.float64)
    eps = 1e-6
    for j in range(Q.shape[1]):
        dp_ = torch.zeros_like(p); dp_[:, j] = eps
        xp = lcp_solve(Q, p + dp_, G, h, A, b, F, **kw).x
        xm = lcp_solve(Q, p - dp_, G, h, A, b, F, **kw).x
        fd[:, j] = (((xp - xm) * cot).sum(1) / (2 * eps)).cpu()
    scale = fd.abs().max(dim=1)[0].clamp_min(1e-12)
    e_adj = ((g_adj - fd).abs().max(dim=1)[0] / scale)
    e_ref = ((g_ref - fd).abs().max(dim=1)[0] / scale)
    # the oracle's transposed solve on the same LCPs
    l64 = [None if t is None else t.cpu() for t in lcp]
    ro = O.lcp_forward(*l64, max_iter=30)
    go = O.lcp_backward(ro, *l64, cot.cpu(), adjoint=True)["dp"]
    e_orc = ((g_adj - go).abs().max(dim=1)[0] / scale)
    print("adjoint vs FD %.2e   reference formula vs FD %.2e   adjoint vs oracle adjoint %.2e" % (float(e_adj.max()), float(e_ref.max()), float(e_orc.max())))
    assert float(e_adj.median()) <= 1e-4 and float(e_orc.median()) <= 1e-6, (e_adj.tolist(), e_orc.tolist())
    assert float(e_ref.max()) > 10 * float(e_adj.median())                                            # the reference's K-solve is NOT the gradient here


@pytest.mark.parametrize("nbox,pts", [(6, 4), (8, 4), (9, 2), (16, 4)])
def test_big_kernel_backward_on_overconverged_solves(nbox, pts):
    """lcp_big.hip's backward (contact space at 17 .. 64 contacts: `path="big"` = LCP_PATH_CONTACT_SPACE, dense class 2) on solves that run
    sixteen iterations - into convergence, s / z of the active rows at 1e-14 and below: the factorisation of lcp.py:46 meets pivots that
    are rounding noise.  Round 6 gave it the second factorisation the other three families have (a pivot below 1e-13 of its row's diagonal of
    W -> s / z floored at 1e-12 x that diagonal): finite gradients on every scene, and the backward at the kernel's own iterate within 1e-5 of
    the oracle's on every scene the oracle says is determined (24, 32, 18 and 64 contacts: the 32- and 64-contact instantiations)."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_backward
    B, ITERS = 256, 16
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=4242 + nbox, dtype=torch.float32)
    lcp = list(O.assemble_lcp(*sc.assembly_args()))
    lcp64 = [None if t is None else t.double() for t in lcp]
    ref = O.lcp_forward(*lcp64, max_iter=ITERS)
    sol = _solve(lcp, torch.float32, path="big", max_iter=ITERS)
    cot = torch.randn(B, lcp64[0].shape[1], generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    grads = lcp_backward(sol, cot.to(device=DEV, dtype=torch.float32))
    g64 = {k: (None if g is None else g.double().cpu()) for k, g in zip("QpGhAbF", grads)}
    assert all(bool(torch.isfinite(g).all()) for g in g64.values() if g is not None)
    fl = parity.grad_floors(lcp64[0], lcp64[1], cot, ref.x, ref.z, ref.y)
    rep = parity.own_iterate_backward(O, lcp64, ref, cot, sol.x.double().cpu(), sol.z.double().cpu(), sol.s.double().cpu(), g64, fl)
    print(nbox, pts, "iters", sol.iters.float().mean().item(), rep)
    # (an over-converged iterate leaves few scenes whose backward system the ORACLE still calls determined - the reference's own backward is
    #  unstable there, profiles/r05_own_iterate_probe.txt - those are compared; on EVERY scene the gradients are finite and no multiplier has
    #  blown up: what a division by a noise pivot returns is dlam ~ 1e15)
    assert float(g64["h"].abs().max()) <= 1e8 * float(cot.abs().max()), float(g64["h"].abs().max())
    if rep["bwd_own_iterate_determined_scenes"] > 0:
        assert rep["bwd_own_iterate_err_max"] <= 1e-5, rep

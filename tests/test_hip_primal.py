"""GPU parity of the body-space PDIPM kernel (`lcp_primal.hip`: the Newton systems of a contact scene solved in nz + neq unknowns,
one wavefront per scene) against the contact-space kernels it replaces behind `lcp_solve_dynamics_f32` /
`lcp_step_backward_f32` (`lcp_big.hip`, forced with the debug path "big") and against the fp64 oracle.  The two formulations
take the same Newton steps in exact arithmetic (pdipm.py:325-454 eliminates x first, the body-space kernel the inequality block),
so new_v has to agree far below the fp32 outputs' resolution."""
import numpy as np
import pytest
import torch

from oracle import pdipm_oracle as O
from tests import parity

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _solve(sc, count, path="auto", pinned=False):
    from lcp_physics_amd import _lib
    from lcp_physics_amd.physics.batched_world import solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    scg = sc.to(device=DEV)
    cb = ContactBuffers(sc.B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    e = 0 if sc.Je is None else sc.Je.shape[1]
    _lib.set_path(path)
    try:
        out = solve_dynamics(sc.B, sc.nb, sc.nc, e, count.to(DEV), scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, scg.Je, sc.dt,
                             pinned=pinned)
        torch.cuda.synchronize()
    finally:
        _lib.set_path("auto")
    return scg, out


def _scenes(kind, B):
    from lcp_physics_amd import scenes
    if kind == "pile":                                    # BASELINE config 5: 11 bodies, 64 contacts -> 36 x 36 systems
        return scenes.make_pile_scenes(B=B, seed=33, dtype=torch.float32)
    nbox, pts = kind
    return scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=500 + 7 * nbox + pts, dtype=torch.float32)


@pytest.mark.parametrize("kind", ["pile", (6, 4), (11, 1), (11, 2), (15, 2), (7, 4)])
def test_forward_matches_the_contact_space_kernels(kind):
    """Piles (system capacity 40), 7 bodies / 24 contacts (24), 12 bodies / 11 and 22 contacts (40), 16 bodies / 30 contacts (56),
    8 bodies / 28 contacts (40), with ragged contact counts: new_v, the iteration counts and the status words of both kernels."""
    B = 48
    sc = _scenes(kind, B)
    g = torch.Generator().manual_seed(3)
    count = torch.randint(0, sc.nc + 1, (B,), generator=g, dtype=torch.int32)
    count[: B // 2] = sc.nc
    _, a = _solve(sc, count)
    _, b = _solve(sc, count, "big")
    va, vb = a["v_new"].double().cpu(), b["v_new"].double().cpu()
    scale = vb.abs().reshape(B, -1).max(dim=1)[0].clamp_min(1.0)
    err = (va - vb).abs().reshape(B, -1).max(dim=1)[0] / scale
    print(kind, "worst scaled |v_new - v_new(contact space)|", float(err.max()), "iteration counts differ in",
          int((a["iters"] != b["iters"]).sum()), "of", B)
    assert float(err.max()) <= 2e-6
    assert int(((a["status"] & 8) != 0).sum()) == 0
    # (iteration counts may differ where the solve converges to rounding - towers with one contact per interface: the exit
    #  tests of pdipm.py:133 are then met a rounding apart; the answers above are the criterion)
    full = (count == sc.nc).to(DEV)
    assert int(((a["iters"] != b["iters"]) & (a["iters"] == 10) & (b["iters"] == 10) & full).sum()) == 0


def test_forward_matches_the_oracle_on_piles():
    B = 8
    sc = _scenes("pile", B)
    count = torch.tensor([64, 64, 40, 64, 9, 64, 64, 23], dtype=torch.int32)
    _, out = _solve(sc, count)
    got = out["v_new"].double().cpu()
    for k in range(B):
        n = int(count[k])
        one = lambda t: t[k:k + 1]
        args = (one(sc.Mdiag), one(sc.v), one(sc.f), sc.dt, sc.c_n[k:k + 1, :n], sc.c_p1[k:k + 1, :n], sc.c_p2[k:k + 1, :n],
                sc.c_i1[k:k + 1, :n], sc.c_i2[k:k + 1, :n], one(sc.rest), one(sc.fric), one(sc.Je))
        lcp64 = [None if t is None else t.double() for t in O.assemble_lcp(*args)]
        rs = O.lcp_forward(*lcp64)
        ex = float(parity.err_x(-got[k].reshape(1, -1), rs.x, lcp64[0], lcp64[1]).max())
        assert ex <= 1e-5, (k, n, ex)


@pytest.mark.parametrize("kind", ["pile", (6, 4), (8, 4), (9, 2), (11, 1), (11, 2), (12, 2)])     # free coordinates: 30 | 18, 24 -> 24 columns | 27 -> 32 | 33, 33, 36 -> 40
def test_pinned_form_matches_the_general_form(kind):
    """LCP_HINT_PINNED on the one-wave-per-scene sizes (`lcp_primal_pin.hip`): the free coordinates' system (nz - neq pivots: 30
    instead of 36 on the piles of BASELINE config 5) against the full KKT system of `lcp_primal.hip` - same iterates up to
    rounding: new_v, z, s, the multipliers y of the pinned rows, iteration counts, status words; ragged contact counts; and the
    physical backward of each from its own forward."""
    from lcp_physics_amd.physics.batched_world import fused_step_backward, rows_pin_leading_coordinates
    B = 48
    sc = _scenes(kind, B)
    assert rows_pin_leading_coordinates(sc.Je)
    g = torch.Generator().manual_seed(3)
    count = torch.randint(0, sc.nc + 1, (B,), generator=g, dtype=torch.int32)
    count[: B // 2] = sc.nc
    scg, a = _solve(sc, count, pinned=True)
    _, b = _solve(sc, count)
    assert (a["compute"] & 0x20000) and not (b["compute"] & 0x20000)
    scale = b["v_new"].double().abs().reshape(B, -1).max(dim=1)[0].clamp_min(1.0).cpu()
    for k, tol in (("v_new", 1e-6), ("y", 1e-5)):
        err = (a[k].double() - b[k].double()).abs().reshape(B, -1).max(dim=1)[0].cpu() / (scale if k == "v_new" else b[k].double().abs().max().cpu().clamp_min(1.0))
        assert float(err.max()) <= tol, (k, float(err.max()))
    # (bit 4, pdipm.py:99-102 "except: return best": an over-converged scene may meet its non-finite pivot in one form and not in
    #  the other - both return the best iterate, compared above)
    assert int((((a["status"] ^ b["status"]) & ~4) != 0).sum()) == 0 and int(((a["status"] | b["status"]) & 8).sum()) == 0
    full = (count == sc.nc).to(DEV)
    d = (a["iters"] - b["iters"]).abs()
    # (truncated lists leave bodies hanging: those solves converge to rounding, where the exit tests of pdipm.py:133 are met a
    #  rounding apart - as in test_forward_matches_the_contact_space_kernels the answers are the criterion there)
    assert int(d.max()) <= 2 and int(d[full].max()) <= 1 and int((d[full] != 0).sum()) <= B // 8, d.tolist()
    same = (d == 0).cpu()
    zs = max(float(b["z"].abs().max()), 1.0)
    assert float((a["z"][same.to(DEV)] - b["z"][same.to(DEV)]).abs().max()) <= 1e-4 * zs
    cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(5), dtype=torch.float32).to(DEV)
    scf = _scenes(kind, B)
    cnt_full = torch.full((B,), sc.nc, dtype=torch.int32)
    scg, a = _solve(scf, cnt_full, pinned=True)
    ga = {k: v.double().cpu() for k, v in fused_step_backward(scg, a, cot).items()}
    scg, b = _solve(scf, cnt_full)
    gb = {k: v.double().cpu() for k, v in fused_step_backward(scg, b, cot).items()}
    for k in ("Mdiag", "v", "f"):
        sc_ = gb[k].abs().reshape(B, -1).max(dim=1)[0]
        sc_ = torch.maximum(sc_, 1e-3 * sc_.max())
        err = (ga[k] - gb[k]).abs().reshape(B, -1).max(dim=1)[0] / sc_
        assert float(err.max()) <= 1e-3, (k, float(err.max()))
    for k in ("rest", "fric", "c_n", "c_p1", "c_p2"):
        assert bool(torch.isfinite(ga[k]).all()), k


def test_pinned_hint_on_other_rows_is_loud():
    """A scene whose equality rows are not [I 0] under LCP_HINT_PINNED is not solved: NaN velocities, LCP_ST_NAN (as in lcp_quad.hip)."""
    B = 8
    sc = _scenes("pile", B)
    sc.Je = sc.Je.clone()
    sc.Je[3, 1, 4] = 0.5
    count = torch.full((B,), sc.nc, dtype=torch.int32)
    _, a = _solve(sc, count, pinned=True)
    bad = torch.isnan(a["v_new"]).reshape(B, -1).any(dim=1).cpu()
    assert bad.tolist() == [k == 3 for k in range(B)]
    assert int(a["status"][3]) & 8 and int((a["status"].cpu()[~bad] & 8).sum()) == 0


def test_results_are_bitwise_reproducible():
    """The matrix and the products G^T w are accumulated with LDS atomics (ds_add_f64) by the one wave that owns the scene: the
    order of the additions is a function of the instruction stream, so two launches give the same bits."""
    sc = _scenes("pile", 256)
    count = torch.full((256,), sc.nc, dtype=torch.int32)
    _, a = _solve(sc, count)
    a = {k: a[k].clone() for k in ("v_new", "z", "s")}
    for _ in range(3):
        _, b = _solve(sc, count)
        for k in a:
            assert torch.equal(a[k].view(torch.int32), b[k].view(torch.int32)), k


@pytest.mark.parametrize("kind", ["pile", (6, 4), (11, 1), (15, 2)])
def test_backward_matches_the_contact_space_backward(kind):
    """`lcp_step_backward_f32` from both kernels on the same scenes (each from its own forward).  (11, 1) is a tower whose solve
    converges to machine precision - the ratios s / z underflow against the masses there, which is what the floored factorisation
    and the refinement step of the body-space backward are for."""
    from lcp_physics_amd import _lib
    from lcp_physics_amd.physics.batched_world import fused_step_backward
    B = 32
    sc = _scenes(kind, B)
    count = torch.full((B,), sc.nc, dtype=torch.int32)
    cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(5), dtype=torch.float32).to(DEV)
    grads = {}
    for path in ("auto", "big"):
        scg, out = _solve(sc, count, path)
        _lib.set_path(path)
        try:
            grads[path] = {k: v.double().cpu() for k, v in fused_step_backward(scg, out, cot).items()}
            torch.cuda.synchronize()
        finally:
            _lib.set_path("auto")
    # Mdiag, v, f: defined whatever the multipliers (tests/parity.py::err_physical); the per-contact keys depend on how the load is
    # shared between redundant contacts, which the two eliminations resolve differently at the level of s / z ~ 1e-12
    worst = 0.0
    for k in ("Mdiag", "v", "f"):
        a, b = grads["auto"][k], grads["big"][k]
        # scale: the scene's largest entry, but not below 1e-3 of the batch's - a stack pinned by friction answers a push with
        # its contact compliances s / z ~ 1e-10 only, and a gradient of 1e-8 next to 1e-2 is zero, not a number to match to 1e-3
        scale = b.abs().reshape(B, -1).max(dim=1)[0]
        scale = torch.maximum(scale, 1e-3 * scale.max())
        err = (a - b).abs().reshape(B, -1).max(dim=1)[0] / scale
        worst = max(worst, float(err.max()))
        assert float(err.max()) <= 1e-3, (k, float(err.max()))
    for k in ("rest", "fric", "c_n", "c_p1", "c_p2"):
        assert bool(torch.isfinite(grads["auto"][k]).all()), k
    print(kind, "worst gradient difference, Mdiag / v / f (relative to the scene's largest entry)", worst)


def test_dense_boundary_routes_every_scene_to_its_kernel():
    """`lcp_pdipm_forward_f32 / _backward_f32` at config-5 sizes: one call serves contact-structured scenes whose rows touch two
    bodies (class 3: lcp_primal.hip; class 4, its pinned form, where A = [I 0] and b = 0), contact-structured scenes that do not (class 2: lcp_big.hip - here a normal row with an entry
    on a third body) and general LCPs (class 0: the generic kernels - here a perturbed F).  The classes the device wrote are read
    back from the tail of the workspace; answers and gradients against the generic kernels forced on every scene."""
    from lcp_physics_amd import _lib, scenes
    from lcp_physics_amd.lcp import lcp_backward, lcp_solve
    B = 12
    sc = scenes.make_pile_scenes(B=B, seed=91, dtype=torch.float32)
    lcp = [None if t is None else t.clone() for t in O.assemble_lcp(*sc.assembly_args())]
    nc, nz, m = sc.nc, 3 * sc.nb, 4 * sc.nc
    lcp[2][2, 5, 3 * 9 + 1] = 0.125                       # scene 2: contact 5 (bodies 0 and 2) also pushes on body 9
    lcp[2][7, 40, 3 * 1 + 2] = -0.25                      # scene 7: likewise
    lcp[6][4, 5, 7] = 0.25                                # scene 4: F is not the contact F any more
    lcp[4][9] = lcp[4][9] * 2.0                           # scene 9: A = 2 [I 0] - the same constraint, but not rows that PIN the coordinates
    g = [None if t is None else t.to(DEV).contiguous() for t in lcp]
    cot = torch.randn(B, nz, generator=torch.Generator().manual_seed(2), dtype=torch.float32).to(DEV)

    def run(path):
        _lib.set_path(path)
        try:
            sol = lcp_solve(*g)
            grads = [None if t is None else t.double().cpu() for t in lcp_backward(sol, cot)]
            torch.cuda.synchronize()
        finally:
            _lib.set_path("auto")
        return sol, grads

    sol, grads = run("auto")
    per_scene = (_lib.workspace_bytes(B, nz, m, 3, _lib.COMPUTE_F64) - ((B * 4 + 255) & ~255) - 256) // B
    cls = sol.ws[B * per_scene: B * per_scene + 4 * B].view(torch.int32).cpu().tolist()
    want = [4] * B                                        # (class 4, round 5: class 3 whose equality rows pin the leading coordinates - lcp_primal_pin.hip)
    want[2] = want[7] = 2
    want[4] = 0
    want[9] = 3
    assert cls == want, cls
    solg, gg = run("generic")
    solb, gb = run("big")                                  # (contact-space kernel on the classes 2 AND 3)
    for other, go, name in ((solg, gg, "generic"), (solb, gb, "big")):
        scale = max(1.0, float(other.x.abs().max()))
        assert float((sol.x - other.x).abs().max()) <= 1e-5 * scale, name
        for k, a, b in zip("QpGhAbF", grads, go):
            if k in "QpAb":                                # (dG, dh, dF of redundant piles are rounding-determined: tests/parity.py)
                sk = float(b.abs().max())
                assert float((a - b).abs().max()) <= 1e-4 * max(sk, 1e-12), (name, k, float((a - b).abs().max()), sk)


def test_quad_body_space_takes_any_equality_rows():
    """The four-scenes-per-wave forward of the contact-list entry points runs a kernel that assumes the equality rows pin the leading
    coordinates (A = [I 0]: the TotalConstraint on the floor) and, behind it, the general body-space kernel for the waves that do
    not qualify.  Here: every fourth wave keeps the floor pin, the others get rows that pin ANOTHER body, a row with general
    entries, or fewer rows - in one batch, so that one call exercises both kernels.  Against the contact-space kernel (forced
    path) and the oracle."""
    from lcp_physics_amd import _lib, scenes
    from lcp_physics_amd.physics import assemble_contacts, fused_step
    B = 64
    sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=17, dtype=torch.float32)
    Je = sc.Je.clone()
    nz = 3 * sc.nb
    for k in range(B):
        kind = (k // 4) % 4                                   # per wave of four scenes
        if kind == 1:                                         # the joint pins body 1 instead of the floor; the floor is heavy
            Je[k].zero_()
            Je[k, 0, 3] = Je[k, 1, 4] = Je[k, 2, 5] = 1.0
        elif kind == 2:                                       # a general row: the floor's x follows body 1's rotation
            Je[k, 1, 3] = 0.25
        elif kind == 3:                                       # scaled rows (same constraint, A is not [I 0])
            Je[k] *= 2.0
    sc.Je = Je
    scg = sc.to(device=DEV)
    out = fused_step(scg)
    _lib.set_path("big")
    try:
        ref_cs = fused_step(scg)
        torch.cuda.synchronize()
    finally:
        _lib.set_path("auto")
    va, vb = out["v_new"].double().cpu(), ref_cs["v_new"].double().cpu()
    scale = vb.abs().reshape(B, -1).max(dim=1)[0].clamp_min(1.0)
    err = (va - vb).abs().reshape(B, -1).max(dim=1)[0] / scale
    assert float(err.max()) <= 2e-6, err.tolist()
    assert int((out["status"] & 8).sum()) == 0
    lcp = [None if t is None else t.double().cpu() for t in assemble_contacts(scg)]
    ref = O.lcp_forward(*lcp)
    ex = parity.err_x(-va.reshape(B, -1), ref.x, lcp[0], lcp[1])
    assert float(ex.max()) <= 1e-4, float(ex.max())
    ey = (out["y"].double().cpu() - ref.y).abs().max(dim=1)[0] / ref.y.abs().max(dim=1)[0].clamp_min(1.0)
    assert float(ey.max()) <= 1e-4, ey.tolist()


def test_quad_body_space_without_equality_rows():
    """neq = 0 (a heavy free floor instead of a pinned one) on the four-scenes-per-wave contact-list path: nothing to pin, the
    pinned-variant kernel factors all nz rows; against the contact-space kernel and the oracle, with ragged contact counts."""
    from lcp_physics_amd import _lib, scenes
    from lcp_physics_amd.physics.batched_world import solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    B = 32
    sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=29, dtype=torch.float32)
    sc.Mdiag[:, 0] *= 1e4
    sc.f[:, 0] = 0
    scg = sc.to(device=DEV)
    cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    count = torch.randint(0, sc.nc + 1, (B,), generator=torch.Generator().manual_seed(4), dtype=torch.int32)
    count[::3] = sc.nc
    run = lambda: solve_dynamics(B, sc.nb, sc.nc, 0, count.to(DEV), scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, None, sc.dt)
    a = run()
    _lib.set_path("big")
    try:
        b = run()
        torch.cuda.synchronize()
    finally:
        _lib.set_path("auto")
    va, vb = a["v_new"].double().cpu(), b["v_new"].double().cpu()
    scale = vb.abs().reshape(B, -1).max(dim=1)[0].clamp_min(1.0)
    assert float(((va - vb).abs().reshape(B, -1).max(dim=1)[0] / scale).max()) <= 2e-6
    assert int((a["status"] & 8).sum()) == 0
    for k in range(0, B, 5):
        n = int(count[k])
        if n == 0:
            continue
        one = lambda t: t[k:k + 1]
        args = (one(sc.Mdiag), one(sc.v), one(sc.f), sc.dt, sc.c_n[k:k + 1, :n], sc.c_p1[k:k + 1, :n], sc.c_p2[k:k + 1, :n],
                sc.c_i1[k:k + 1, :n], sc.c_i2[k:k + 1, :n], one(sc.rest), one(sc.fric), None)
        lcp64 = [None if t is None else t.double() for t in O.assemble_lcp(*args)]
        rs = O.lcp_forward(*lcp64)
        ex = float(parity.err_x(-va[k].reshape(1, -1), rs.x, lcp64[0], lcp64[1]).max())
        assert ex <= 1e-4, (k, n, ex)


@pytest.mark.parametrize("nbox,pts", [(4, 4), (6, 4), (11, 2)])
def test_post_stabilization_body_space_matches_the_generic_kernel_and_the_oracle(nbox, pts):
    """`lcp_post_stabilization_f32` (engines.py:80-116 + the correction move of world.py:109-117) on the body-space kernel
    against the generic workgroup-per-scene kernel (forced path) and the oracle: ragged contact counts (0 = the direct KKT solve),
    perturbed velocities so that the contacts have something to correct."""
    from lcp_physics_amd import _lib, scenes
    from lcp_physics_amd.physics.batched_world import post_stabilization
    from lcp_physics_amd.physics.contacts import ContactBuffers
    from oracle import world_oracle as WO
    B = 48
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=300 + nbox, dtype=torch.float32)
    sc.v = sc.v + 0.3 * torch.randn(sc.v.shape, generator=torch.Generator().manual_seed(1))
    scg = sc.to(device=DEV)
    cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    count = torch.randint(0, sc.nc + 1, (B,), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    count[:8] = sc.nc
    p = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(3), dtype=torch.float64).to(DEV)
    dts = (0.01 + 0.02 * torch.rand(B, generator=torch.Generator().manual_seed(4), dtype=torch.float64)).to(DEV)
    pa, pb = torch.empty_like(p), torch.empty_like(p)
    run = lambda po: post_stabilization(B, sc.nb, sc.nc, 3, count.to(DEV), scg.Mdiag, scg.v, scg.rest, cb, scg.Je, p=p, dt_scene=dts, p_out=po)
    a = run(pa)
    _lib.set_path("generic")
    try:
        b = run(pb)
        torch.cuda.synchronize()
    finally:
        _lib.set_path("auto")
    move = p + (a["dp"].double() * 0.5) * dts.reshape(B, 1, 1)           # world.py:110-117 (the kernel moves with its fp64 dp, `dp` is its fp32 copy)
    assert float((pa - move).abs().max()) <= 1e-7 and float((pa - pb).abs().max()) <= 1e-5
    da, db = a["dp"].double().cpu(), b["dp"].double().cpu()
    scale = db.abs().reshape(B, -1).max(dim=1)[0].clamp_min(1.0)
    err = (da - db).abs().reshape(B, -1).max(dim=1)[0] / scale
    print("post-stabilisation, body space vs generic: worst scaled |dp - dp'|", float(err.max()))
    assert float(err.max()) <= 1e-5
    assert int((a["status"] & 8).sum()) == 0
    for k in range(0, B, 7):                                  # the oracle, scene by scene (engines.py:80-116 restated)
        n = int(count[k])
        contacts = [((sc.c_n[k, c].numpy(), sc.c_p1[k, c].numpy(), sc.c_p2[k, c].numpy(), 0.0), int(sc.c_i1[k, c]), int(sc.c_i2[k, c])) for c in range(n)]
        ref = WO.post_stabilization(sc.Mdiag[k].numpy(), sc.v[k].numpy(), contacts, sc.rest[k].numpy(), sc.Je[k].numpy())
        e = float(np.abs(da[k].numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
        assert e <= 2e-3, (k, n, e)                           # (the bound of the trajectory test: this LCP is ill-conditioned at rest)


@pytest.mark.parametrize("nbox,pts,rows", [(4, 4, "pinned"), (2, 4, "pinned"), (4, 2, "moving_floor"), (3, 4, "scaled")])
def test_post_stabilization_four_scenes_per_wave_matches_one_wave_per_scene(nbox, pts, rows):
    """Round 5: at nz <= 16, <= 16 contacts `lcp_post_stabilization_f32` runs `lcp_fwd_quad<..., POST>` (four scenes per wavefront; the
    pinned body-space kernel, the general one behind it) where it ran `lcp_poststab_primal_kernel` (one wavefront per scene:
    LCP_PATH_PRIMAL keeps it).  Same LCP (engines.py:80-116), same correction move, and the SAME workspace layout: the one backward kernel
    (`lcp_post_stabilization_backward_f32`) follows either forward.  `moving_floor`: the pinned body has a velocity, so b = Je v is not zero
    and no wave qualifies for the pinned kernel; `scaled`: A = 2 [I 0] - not rows that pin."""
    from lcp_physics_amd import _lib, scenes
    from lcp_physics_amd.physics.batched_world import post_stabilization, post_stabilization_backward
    from lcp_physics_amd.physics.contacts import ContactBuffers
    B = 203                                                   # (a last wavefront with three live scenes)
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=500 + nbox + pts, dtype=torch.float32)
    sc.v = sc.v + 0.3 * torch.randn(sc.v.shape, generator=torch.Generator().manual_seed(1))
    if rows == "moving_floor":
        sc.v[:, 0] = 0.05 * torch.randn(B, 3, generator=torch.Generator().manual_seed(5))
    elif rows == "scaled":
        sc.Je = sc.Je * 2.0
    else:
        sc.v[:, 0] = 0.0
    scg = sc.to(device=DEV)
    cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    count = torch.randint(0, sc.nc + 1, (B,), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    count[:64] = sc.nc
    p = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(3), dtype=torch.float64).to(DEV)
    dts = (0.01 + 0.02 * torch.rand(B, generator=torch.Generator().manual_seed(4), dtype=torch.float64)).to(DEV)
    cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(6), dtype=torch.float32).to(DEV)
    res = {}
    for path in ("auto", "primal"):
        _lib.set_path(path)
        try:
            po = torch.empty_like(p)
            out = post_stabilization(B, sc.nb, sc.nc, 3, count.to(DEV), scg.Mdiag, scg.v, scg.rest, cb, scg.Je, p=p, dt_scene=dts, p_out=po)
            g = post_stabilization_backward(B, sc.nb, sc.nc, 3, scg.Mdiag, scg.v, scg.rest, cb, scg.Je, cot, out, want_Je=True)
            torch.cuda.synchronize()
        finally:
            _lib.set_path("auto")
        res[path] = (out["dp"].double().cpu(), po.cpu(), out["iters"].cpu(), out["status"].cpu(), {k: v.double().cpu() for k, v in g.items()})
    a, b = res["auto"], res["primal"]
    scale = b[0].abs().reshape(B, -1).max(dim=1)[0].clamp_min(1.0)
    err = (a[0] - b[0]).abs().reshape(B, -1).max(dim=1)[0] / scale
    assert float(err.max()) <= 1e-6, float(err.max())
    assert float((a[1] - b[1]).abs().max()) <= 1e-7                       # the moved poses (fp64)
    assert int(((a[3] | b[3]) & 8).sum()) == 0                            # no NaN status either way
    # (solves of a few contacts converge to rounding inside the ten iterations, the exit tests of pdipm.py:133 then compare rounding noise:
    #  the two eliminations can leave such a solve a few passes apart - dp above is unaffected)
    assert int((a[2] - b[2]).abs().max()) <= 4
    # the backward read either forward's workspace: gradients agree where the two forwards kept the same iterate
    same = (a[2] == b[2])
    assert int(same.sum()) >= int(0.6 * B)
    for k in ("Mdiag", "v", "rest", "Je"):
        if k not in a[4]:
            continue
        ga, gb = a[4][k].reshape(B, -1)[same], b[4][k].reshape(B, -1)[same]
        den = gb.abs().max(dim=1)[0].clamp_min(1e-6 * float(gb.abs().max()))
        assert bool(torch.isfinite(ga).all()), k
        assert float(((ga - gb).abs().max(dim=1)[0] / den).max()) <= 1e-3, (k, float(((ga - gb).abs().max(dim=1)[0] / den).max()))


def _with_joint_rows(sc, e_target):
    """Stack scenes with more joints: next to the TotalConstraint on the floor (3 rows) the top box is welded to the world (3),
    then X / Rot / Y constraints on the boxes in between, until `e_target` rows (constraints.py:95-217: rows of the identity) -
    and, so that the Jacobian is not only unit rows, the last row is a revolute-style row coupling two bodies."""
    B, nb = sc.B, sc.nb
    nz = 3 * nb
    rows = []
    unit = lambda c: torch.zeros(nz).index_fill_(0, torch.tensor([c]), 1.0)
    for q in range(3):
        rows.append(unit(3 * (nb - 1) + q))
    for b in range(1, nb - 1):
        for q in (1, 0, 2):
            rows.append(unit(3 * b + q))
    rows = rows[: e_target - 3]
    r = rows[-1].clone()                                       # couple the last constrained coordinate to the floor's (pinned) rotation
    r[0] = -12.5
    rows[-1] = r
    extra = torch.stack(rows).to(sc.Je.dtype).unsqueeze(0).repeat(B, 1, 1)
    sc.Je = torch.cat([sc.Je, extra], dim=1).contiguous()
    assert sc.Je.shape[1] == e_target
    return sc


@pytest.mark.parametrize("nbox,pts,e", [(4, 2, 7), (4, 4, 12), (8, 2, 12), (8, 2, 16), (12, 2, 16), (8, 2, 24), (10, 2, 20), (11, 2, 20)])
def test_chains_of_joints_up_to_24_equality_rows(nbox, pts, e):
    """5 .. 24 equality rows (a chain of revolute joints has two per link: the reference's chain demo, `testChain`, the ten links
    of experiments/inference.py) run on the body-space kernel's 24-row instantiation, forward and backward: new_v, y against the generic kernel and the oracle; the
    gradients of `lcp_step_backward_je_f32` (Mdiag, v, f and dJe) against the generic kernels' dense backward (lcp.py:52-61)
    contracted through the assembly."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_backward
    from lcp_physics_amd.physics.batched_world import assemble_contacts, fused_step, fused_step_backward, solution_of_step
    B = 32
    sc = _with_joint_rows(scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=900 + nbox + e, dtype=torch.float32), e)
    count = torch.full((B,), sc.nc, dtype=torch.int32)
    scg, out = _solve(sc, count)
    gen = fused_step(scg, path="generic")
    torch.cuda.synchronize()
    va, vb = out["v_new"].double().cpu(), gen["v_new"].double().cpu()
    scale = vb.abs().reshape(B, -1).max(dim=1)[0].clamp_min(1.0)
    err = (va - vb).abs().reshape(B, -1).max(dim=1)[0] / scale
    assert float(err.max()) <= 2e-6, err.tolist()
    assert int((out["status"] & 8).sum()) == 0
    lcp = [None if t is None else t.double().cpu() for t in assemble_contacts(scg)]
    ref = O.lcp_forward(*lcp)
    ex = parity.err_x(-va.reshape(B, -1), ref.x, lcp[0], lcp[1])
    assert float(ex.max()) <= 1e-4, float(ex.max())
    ey = (out["y"].double().cpu() - ref.y).abs().max(dim=1)[0] / ref.y.abs().max(dim=1)[0].clamp_min(1.0)
    assert float(ey.max()) <= 1e-4, ey.tolist()
    # backward
    cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(21), dtype=torch.float32)
    pg = {k: v.double().cpu() for k, v in fused_step_backward(scg, out, cot.to(DEV), want_Je=True).items()}
    lcpd = assemble_contacts(scg)
    dense = lcp_backward(solution_of_step(scg, gen, lcpd[2], lcpd[4]), (-cot).reshape(B, -1).to(DEV))
    torch.cuda.synchronize()
    dense = {k: (None if t is None else t.double().cpu()) for k, t in zip("QpGhAbF", dense)}
    ph = {k: v.double() if v.is_floating_point() else v for k, v in sc.phys_dict().items()}
    refg = parity.physical_grads(ph, sc.dt, dense, O)
    for k in ("Mdiag", "v", "f"):
        scale = refg[k].abs().reshape(B, -1).max(dim=1)[0].clamp_min(1e-30)
        eg = (pg[k] - refg[k]).abs().reshape(B, -1).max(dim=1)[0] / scale
        big = scale > 1e-6 * scale.max()
        assert float(eg[big].max()) < 2e-3, (k, float(eg[big].max()), int(eg.argmax()))
    scale = dense["A"].abs().reshape(B, -1).max(dim=1)[0].clamp_min(1e-30)
    eg = (pg["Je"] - dense["A"]).abs().reshape(B, -1).max(dim=1)[0] / scale
    assert float(eg.max()) < 2e-3, ("Je", float(eg.max()), int(eg.argmax()))

"""GPU: differentiable steps beyond the sizes of rounds 1-5's one-wave kernels (3 nb + e > 56, > 64 contacts, fp32 arithmetic beyond 16
contacts).  Since round 6 `SolveDynamicsFunction` keeps them on the device: 18 .. 20 bodies (3 nb + e <= 64) run on the 64-row
instantiation of the body-space kernel, everything beyond on the generic kernels - `lcp_step_kernel` leaves its iterate and
`lcp_step_bwd_kernel` (lcp_generic.hip) contracts `lcp.py:37-64` through the assembly (here: more than 64 contacts, fp32 arithmetic,
a forced generic path) - no host synchronisation, no RuntimeWarning.  The dense boundary
(`lcp_physics_amd/physics/dense_step.py`: torch assembly on the device + `LCPFunction`, the reference's own route,
`engines.py:26-116`, `lcp.py:20-64`) remains for what still has no fused backward (the wave64 step family: fp32 arithmetic, 3 nb <= 16
with 5..8 joint rows) and is tested here on the same scenes beside the fused route.  Against the fp64 oracle end to end (forward, and
`lcp.py:37-64` + autograd through the assembly for every physical gradient), against the fused no-grad step of the same scenes, and
through `ContactWorld`."""
import pytest
import torch

from oracle import pdipm_oracle as O
from tests import parity

pytestmark = pytest.mark.gpu
DEV = "cuda"
KEYS = ("Mdiag", "v", "f", "rest", "fric", "c_n", "c_p1", "c_p2")


def _tall_stack(B=6, nbox=19, pts=2, seed=11):
    from lcp_physics_amd import scenes
    return scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=seed, dtype=torch.float32)


def _leaves(sc):
    return {k: getattr(sc, k).to(DEV).clone().requires_grad_(True) for k in KEYS}


def _oracle_step(sc, cot, count=None):
    """fp64 oracle per scene with its own contact count: v_new and the physical gradients of sum(v_new * cot)."""
    vs, gs = [], {k: [] for k in KEYS}
    for i in range(sc.B):
        one = sc.slice(i, i + 1)
        nc = sc.nc if count is None else int(count[i])
        L = {k: getattr(one, k).double().clone().requires_grad_(True) for k in KEYS}
        cut = lambda t: t[:, :nc]
        if nc == 0:
            Md = L["Mdiag"].reshape(1, -1)
            top = Md * L["v"].reshape(1, -1) + sc.dt * L["f"].reshape(1, -1)
            Je = one.Je.double()
            P = torch.cat([torch.cat([torch.diag_embed(Md), -Je.transpose(1, 2)], 2),
                           torch.cat([Je, torch.zeros(1, Je.shape[1], Je.shape[1], dtype=torch.float64)], 2)], 1)
            v_new = (torch.inverse(P) @ torch.cat([top, torch.zeros(1, Je.shape[1], dtype=torch.float64)], 1).unsqueeze(2))
            v_new = v_new.squeeze(2)[:, :Md.shape[1]].reshape(1, sc.nb, 3)                     # engines.py:36-49
            g = torch.autograd.grad((v_new * cot[i:i + 1].double()).sum(), [L[k] for k in KEYS], allow_unused=True)
        else:
            lcp = O.assemble_lcp(L["Mdiag"], L["v"], L["f"], sc.dt, cut(L["c_n"]), cut(L["c_p1"]), cut(L["c_p2"]), cut(one.c_i1),
                                 cut(one.c_i2), L["rest"], L["fric"], one.Je.double())
            det = [None if t is None else t.detach() for t in lcp]
            sol = O.lcp_forward(*det)
            v_new = (-sol.x).reshape(1, sc.nb, 3)
            gl = O.lcp_backward(sol, *det, (-cot[i:i + 1].double()).reshape(1, -1))          # lcp.py:37-64
            outs = [t for t, k in zip(lcp, "QpGhAbF") if k in "QpGhF"]
            g = torch.autograd.grad(outs, [L[k] for k in KEYS], [gl["d" + k] for k in "QpGhF"], allow_unused=True)
        vs.append(v_new.detach())
        for k, gi in zip(KEYS, g):
            gs[k].append(torch.zeros_like(L[k]) if gi is None else gi)
    return torch.cat(vs), {k: torch.cat(v) for k, v in gs.items()}


def _rel(a, b):
    B = a.shape[0]
    return ((a - b).abs().reshape(B, -1).max(dim=1)[0] / b.abs().reshape(B, -1).max(dim=1)[0].clamp_min(1e-30))


def _recorded_step(route, sc, L, scg, count, Je, opts):
    """route "fused": `SolveDynamicsFunction` (which must stay on the device kernels); "dense": the dense boundary called directly."""
    from lcp_physics_amd.physics.batched_world import SolveDynamicsFunction
    from lcp_physics_amd.physics.dense_step import solve_dynamics_dense
    fn = SolveDynamicsFunction.apply if route == "fused" else solve_dynamics_dense
    import warnings
    with warnings.catch_warnings():
        if route == "fused":
            warnings.simplefilter("error", RuntimeWarning)
        v_new = fn(L["Mdiag"], L["v"], L["f"], L["rest"], L["fric"], L["c_n"], L["c_p1"], L["c_p2"], scg.c_i1, scg.c_i2, count, Je, sc.dt, opts)
    assert bool(opts["last"].get("dense_boundary")) == (route == "dense")
    return v_new


@pytest.mark.parametrize("route", ["fused", "dense"])
@pytest.mark.parametrize("pts", [1, 2, 3])
def test_a_recorded_step_of_twenty_bodies(pts, route):
    """20 bodies (3 nb + e = 63: since round 6 the 64-row instantiation of the body-space kernel, one wavefront per scene); 19 contacts
    (one point per interface: a non-redundant contact set, every gradient determined), 38 and 57 contacts (for the dense route: nineq 228,
    the matrices no longer fit the 160 KB of LDS in fp64 - the generic kernels' workspace plan, lcp_generic.hip carve() level 2)."""
    from lcp_physics_amd import _lib
    from lcp_physics_amd.physics.batched_world import fused_step
    sc = _tall_stack(pts=pts, B=3 if pts == 3 else 6)
    assert _lib.load().lcp_step_has_backward(sc.nb, sc.nc, 3, _lib.COMPUTE_F64)
    L = _leaves(sc)
    scg = sc.to(device=DEV)
    opts = {"max_iter": 10, "compute": "f64"}
    v_new = _recorded_step(route, sc, L, scg, None, scg.Je, opts)
    assert v_new.dtype == torch.float32 and v_new.shape == (sc.B, sc.nb, 3)
    cot = torch.randn(sc.B, sc.nb, 3, generator=torch.Generator().manual_seed(4))
    (v_new * cot.to(DEV)).sum().backward()
    v_ref, g_ref = _oracle_step(sc, cot)
    # forward: the oracle, and the fused (no-grad) step of the same scenes - the value must not depend on whether a step is recorded
    assert float(_rel(v_new.detach().double().cpu(), v_ref).max()) < 2e-6
    fused = fused_step(scg)["v_new"].double().cpu()
    assert float(_rel(v_new.detach().double().cpu(), fused).max()) < 2e-6
    assert torch.equal(opts["last"]["iters"].cpu(), fused_step(scg)["iters"].cpu())
    # backward: every physical gradient against lcp.py:37-64 + autograd in fp64
    worst = {k: float(_rel(L[k].grad.double().cpu(), g_ref[k]).max()) for k in KEYS}
    print(route, "step, worst relative gradient error per key:", worst)
    for k in ("Mdiag", "v", "f"):
        assert worst[k] < 1e-4, (k, worst)
    # The frame / restitution / friction gradients are functions of the multipliers' sensitivities, and two points per interface are
    # already redundant in the tangential direction (DESIGN round 5, item 2: the split of the friction gradient between the two points is
    # left to rounding, the sum is determined).  The dense route factors the oracle's own contact-space system and reproduces its split;
    # the fused route (round 6: the body-space kernel up to 64 rows) has its own - there the per-interface SUMS are compared.
    if pts == 1 or (pts == 2 and route == "dense"):
        # (friction: where nothing slides the cone multipliers and their sensitivities are rounding zeros - compared only where the
        #  oracle's own gradient is not noise against the restitution's)
        for k in ("rest", "c_n", "c_p1", "c_p2"):
            assert worst[k] < 1e-3, (k, worst)
        # friction: d(loss)/d(mu) = -dlam_gamma lam_n (lcp.py:54) hangs on the cone multiplier's sensitivity, which a contact sitting ON the
        # cone leaves to rounding (measured: on one scene of six both routes differ from the oracle, by 10 % and by 100 %, while every other
        # gradient of that scene agrees to 1e-6) - the typical scene is compared
        ef = _rel(L["fric"].grad.double().cpu(), g_ref["fric"])
        print("   friction gradient per scene:", ef.tolist())
        assert float(ef.median()) < 1e-3, ef
    if pts == 2 and route == "fused":
        pair = lambda t: t.reshape(sc.B, sc.nc // 2, 2, 2).sum(dim=2)
        sums = {k: float(_rel(pair(L[k].grad.double().cpu()), pair(g_ref[k])).max()) for k in ("c_n", "c_p1", "c_p2")}
        print("fused step, per-interface sums of the frame gradients:", sums, "rest / fric:", worst["rest"], worst["fric"])
        assert worst["rest"] < 1e-5 and worst["fric"] < 1e-2, worst
        assert sums["c_p1"] < 1e-5 and sums["c_p2"] < 1e-5, sums             # (the arms' sums are determined; the normals' are not)


@pytest.mark.parametrize("route", ["fused", "dense"])
def test_scenes_with_their_own_contact_counts_are_solved_with_exactly_those_contacts(route):
    from lcp_physics_amd.physics.batched_world import solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    sc = _tall_stack(B=5, seed=12)
    count = torch.tensor([sc.nc, 0, 20, 20, sc.nc + 3], dtype=torch.int32)
    L = _leaves(sc)
    scg = sc.to(device=DEV)
    opts = {"max_iter": 10, "compute": "f64"}
    v_new = _recorded_step(route, sc, L, scg, count.to(DEV), scg.Je, opts)
    cot = torch.randn(sc.B, sc.nb, 3, generator=torch.Generator().manual_seed(5))
    (v_new * cot.to(DEV)).sum().backward()
    v_ref, g_ref = _oracle_step(sc, cot, count.clamp(max=sc.nc))
    assert float(_rel(v_new.detach().double().cpu(), v_ref).max()) < 2e-6
    from lcp_physics_amd import _lib
    assert (opts["last"]["status"].cpu() & _lib.ST_TRUNCATED).bool().tolist() == [False, False, False, False, True]
    # the fused step with the same counts (no grad): same velocities, same multipliers in the same row layout
    cb = ContactBuffers(sc.B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    out = solve_dynamics(sc.B, sc.nb, sc.nc, 3, count.to(DEV), scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, scg.Je, sc.dt)
    assert float(_rel(v_new.detach().double().cpu(), out["v_new"].double().cpu()).max()) < 2e-6
    zs = out["z"].double().cpu().abs().max(dim=1, keepdim=True)[0].clamp_min(1e-30)
    assert float(((opts["last"]["z"].double().cpu() - out["z"].double().cpu()).abs() / zs).max()) < 1e-4
    for k in ("Mdiag", "v", "f"):
        assert float(_rel(L[k].grad.double().cpu(), g_ref[k]).max()) < 1e-4, k
    # padded contact slots receive no gradient
    for i, c in enumerate(count.clamp(max=sc.nc).tolist()):
        for k in ("c_n", "c_p1", "c_p2"):
            assert float(L[k].grad[i, c:].abs().max() if c < sc.nc else 0.0) == 0.0


@pytest.mark.parametrize("route", ["fused", "dense"])
def test_post_stabilization_of_twenty_bodies(route):
    import warnings
    from lcp_physics_amd import _lib
    from lcp_physics_amd.physics.batched_world import PostStabilizationFunction, post_stabilization
    from lcp_physics_amd.physics.contacts import ContactBuffers
    from lcp_physics_amd.physics.dense_step import post_stabilization_dense
    sc = _tall_stack(B=4, seed=13)
    assert _lib.load().lcp_post_stabilization_has_backward(sc.nb, sc.nc, 3, _lib.COMPUTE_F64)     # (round 6: lcp_step_bwd_kernel<.., POST>)
    assert _lib.load().lcp_post_stabilization_has_backward(5, 16, 3, _lib.COMPUTE_F64)
    keys = ("Mdiag", "v", "rest", "c_n", "c_p1", "c_p2")
    L = {k: getattr(sc, k).to(DEV).clone().requires_grad_(True) for k in keys}
    scg = sc.to(device=DEV)
    opts = {"max_iter": 10, "compute": "f64"}
    fn = PostStabilizationFunction.apply if route == "fused" else post_stabilization_dense
    with warnings.catch_warnings():
        if route == "fused":
            warnings.simplefilter("error", RuntimeWarning)
        dp = fn(L["Mdiag"], L["v"], L["rest"], L["c_n"], L["c_p1"], L["c_p2"], scg.c_i1, scg.c_i2, None, scg.Je, opts)
    cot = torch.randn(sc.B, sc.nb, 3, generator=torch.Generator().manual_seed(6))
    (dp * cot.to(DEV)).sum().backward()
    # oracle: engines.py:80-116 + lcp.py:37-64 + autograd
    R = {k: getattr(sc, k).double().clone().requires_grad_(True) for k in keys}
    lcp = O.assemble_post_stabilization(R["Mdiag"], R["v"], R["c_n"], R["c_p1"], R["c_p2"], sc.c_i1, sc.c_i2, R["rest"], sc.Je.double())
    det = [None if t is None else t.detach() for t in lcp]
    sol = O.lcp_forward(*det)
    gl = O.lcp_backward(sol, *det, (-cot.double()).reshape(sc.B, -1))
    pairs = [(t, gl["d" + k]) for t, k in zip(lcp, "QpGhAbF") if t is not None and t.requires_grad]
    g_ref = torch.autograd.grad([t for t, _ in pairs], [R[k] for k in keys], [g for _, g in pairs], allow_unused=True)
    assert float(_rel(dp.detach().double().cpu(), (-sol.x).reshape(sc.B, sc.nb, 3)).max()) < 1e-5
    cb = ContactBuffers(sc.B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    full = torch.full((sc.B,), sc.nc, dtype=torch.int32, device=DEV)
    fused = post_stabilization(sc.B, sc.nb, sc.nc, 3, full, scg.Mdiag, scg.v, scg.rest, cb, scg.Je)["dp"].double().cpu()
    assert float(_rel(dp.detach().double().cpu(), fused).max()) < 1e-5
    worst = {k: float(_rel(L[k].grad.double().cpu(), torch.zeros_like(R[k]) if g is None else g).max()) for k, g in zip(keys, g_ref)}
    print(route, "post-stabilisation, worst relative gradient error per key:", worst)
    for k in ("Mdiag", "v"):
        assert worst[k] < 1e-3, (k, worst)
    if route == "fused":                      # (two points per interface: unique multipliers, so the frame and restitution gradients are too)
        for k in ("rest", "c_n", "c_p1", "c_p2"):
            assert worst[k] < 1e-2, (k, worst)


@pytest.mark.parametrize("post_stab", [False, True])
def test_contact_world_records_steps_of_twenty_bodies(post_stab):
    """`ContactWorld.step(differentiable=True)` with 20 bodies (3 nb + e = 63): the recorded roll-out follows the plain one, and a loss
    on the final pose reaches the initial velocities and the masses - on the device kernels alone: no RuntimeWarning (the dense
    boundary's), and no synchronising torch call in the steps or in the backward (`torch.cuda.set_sync_debug_mode("error")`)."""
    import warnings
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics.batched_world import ContactWorld
    from lcp_physics_amd.physics.contacts import GeometryBatch
    B, nbox, steps = 4, 19, 4
    w = scenes.make_drop_world(B, nbox=nbox, seed=77, gap=(0.02, 0.08))        # (inside the detection margin: in contact from the start)
    geom = GeometryBatch.from_shapes(w["shapes"], B).to(DEV)
    dev = lambda t: t.to(DEV)

    def world(Mdiag, v, p=None):
        return ContactWorld(geom, dev(w["p"]) if p is None else p, v, Mdiag, dev(w["f"]), dev(w["rest"]), dev(w["fric"]), Je=dev(w["Je"]), maxc=48,
                            post_stab=post_stab)

    plain = world(dev(w["Mdiag"]), dev(w["v"]))
    for _ in range(steps):
        plain.step()
    Mdiag, v0, p0 = dev(w["Mdiag"]).requires_grad_(True), dev(w["v"]).requires_grad_(True), dev(w["p"]).requires_grad_(True)
    rec = world(Mdiag, v0, p0)
    torch.cuda.synchronize()
    mode = torch.cuda.get_sync_debug_mode()
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)
            torch.cuda.set_sync_debug_mode("error")
            for _ in range(steps):
                rec.step(differentiable=True)
            loss = (rec.p[:, 1:, 1:] ** 2).sum() * 1e-4
            loss.backward()
    finally:
        torch.cuda.set_sync_debug_mode(mode)
    assert float((rec.p.detach() - plain.p).abs().max()) < 1e-4
    assert int(rec.contacts.count.max()) > 16
    for t in (Mdiag.grad, v0.grad, p0.grad):                   # (p0: through the contact frames of 20 bodies - lcp_contact_frame_backward_f64)
        assert t is not None and bool(torch.isfinite(t).all()) and float(t.abs().max()) > 0


@pytest.mark.parametrize("route", ["fused", "dense"])
@pytest.mark.parametrize("compute,with_joints", [("f64", False), ("f32", True)])
def test_recorded_steps_without_joints_and_in_fp32_arithmetic(compute, with_joints, route):
    """The same two routes with no equality rows at all (A, b empty - `engines.py:59-60`) and in fp32 arithmetic (`compute="f32"`: beyond
    16 contacts the generic kernels at any number of bodies): forward against the fused step of the same word, gradients against the
    fp64 oracle at the tolerance of the arithmetic."""
    from lcp_physics_amd import _lib
    from lcp_physics_amd.physics.batched_world import solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    sc = _tall_stack(B=4, seed=15) if compute == "f64" else _tall_stack(B=4, nbox=6, pts=4, seed=16)     # (fp32: 7 bodies, 24 contacts)
    e = 3 if with_joints else 0
    assert _lib.load().lcp_step_has_backward(sc.nb, sc.nc, e, _lib.COMPUTE_F64 if compute == "f64" else _lib.COMPUTE_F32)
    L = _leaves(sc)
    scg = sc.to(device=DEV)
    Je = scg.Je if with_joints else None
    opts = {"max_iter": 10, "compute": compute}
    v_new = _recorded_step(route, sc, L, scg, None, Je, opts)
    cot = torch.randn(sc.B, sc.nb, 3, generator=torch.Generator().manual_seed(8))
    (v_new * cot.to(DEV)).sum().backward()
    cb = ContactBuffers(sc.B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    full = torch.full((sc.B,), sc.nc, dtype=torch.int32, device=DEV)
    fused = solve_dynamics(sc.B, sc.nb, sc.nc, e, full, scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, Je, sc.dt, compute=compute)
    # (fp32 arithmetic on four collinear points per interface: the two routes round the assembly differently and the solve amplifies it)
    tol = 2e-6 if compute == "f64" else 2e-2
    assert float(_rel(v_new.detach().double().cpu(), fused["v_new"].double().cpu()).max()) < tol
    import dataclasses
    ref_sc = sc if with_joints else dataclasses.replace(sc, Je=torch.zeros(sc.B, 0, 3 * sc.nb))
    v_ref, g_ref = _oracle_step(ref_sc, cot)
    assert float(_rel(v_new.detach().double().cpu(), v_ref).max()) < tol
    for k in ("Mdiag", "v", "f"):
        err = _rel(L[k].grad.double().cpu(), g_ref[k])
        assert float(err.median()) < (1e-4 if compute == "f64" else 5e-2), (k, err)


def test_the_wave64_step_family_still_takes_the_dense_boundary_and_says_so():
    """3 nb <= 16 with 5..8 joint rows in fp32 arithmetic (lcp_wave64.hip's step; in fp64 the body-space kernels take these sizes)
    keeps no iterate a fused backward could read: a recorded step of that shape goes through the dense boundary, with its
    RuntimeWarning, and agrees with the oracle at the tolerance of the arithmetic."""
    import dataclasses
    from lcp_physics_amd import _lib
    from lcp_physics_amd.physics import dense_step
    from lcp_physics_amd.physics.batched_world import SolveDynamicsFunction
    sc = _tall_stack(B=4, nbox=4, pts=2, seed=21)                               # 5 bodies, 3 nb = 15
    Je = torch.zeros(sc.B, 6, 3 * sc.nb)
    Je[:, :3, :3] = torch.eye(3)                                                # the ground pinned (as the scenes have it) ...
    Je[:, 3, 3 * (sc.nb - 1) + 0] = 1.0                                         # ... and the top box: no turning, no sideways motion,
    Je[:, 4, 3 * (sc.nb - 1) + 1] = 1.0                                         #     no vertical motion
    Je[:, 5, 3 * (sc.nb - 1) + 2] = 1.0
    sc = dataclasses.replace(sc, Je=Je)
    assert sc.nc <= 16 and not _lib.load().lcp_step_has_backward(sc.nb, sc.nc, 6, _lib.COMPUTE_F32)
    assert _lib.load().lcp_step_has_backward(sc.nb, sc.nc, 6, _lib.COMPUTE_F64)
    L = _leaves(sc)
    scg = sc.to(device=DEV)
    opts = {"max_iter": 10, "compute": "f32"}
    dense_step._WARNED.clear()
    with pytest.warns(RuntimeWarning, match="dense LCPFunction boundary"):
        v_new = SolveDynamicsFunction.apply(L["Mdiag"], L["v"], L["f"], L["rest"], L["fric"], L["c_n"], L["c_p1"], L["c_p2"], scg.c_i1,
                                            scg.c_i2, None, scg.Je, sc.dt, opts)
    assert opts["last"]["dense_boundary"]
    cot = torch.randn(sc.B, sc.nb, 3, generator=torch.Generator().manual_seed(9))
    (v_new * cot.to(DEV)).sum().backward()
    v_ref, g_ref = _oracle_step(sc, cot)
    assert float(_rel(v_new.detach().double().cpu(), v_ref).max()) < 1e-3
    for k in ("Mdiag", "v", "f"):
        assert float(_rel(L[k].grad.double().cpu(), g_ref[k]).median()) < 5e-2, k


@pytest.mark.parametrize("compute,nbox,pts", [("f64", 33, 2), ("f64", 11, 6), ("f32", 11, 6)])
def test_a_recorded_step_beyond_64_contacts(compute, nbox, pts):
    """66 contacts: 34 bodies with two points per interface (unique multipliers: the tight tolerances), and 12 bodies with six collinear
    points per interface (3 nb + e = 39 would fit the one-wave kernels, the contact count does not; the contact set is redundant four times
    over, so the solve is degenerate and the tolerances are those of the degeneracy).  The generic kernels in both directions, each scene
    at its own count."""
    from lcp_physics_amd import _lib
    sc = _tall_stack(B=4, nbox=nbox, pts=pts, seed=31)
    assert sc.nc > 64
    comp = _lib.COMPUTE_F64 if compute == "f64" else _lib.COMPUTE_F32
    assert _lib.load().lcp_step_has_backward(sc.nb, sc.nc, 3, comp)
    count = torch.tensor([sc.nc, 64, 30, 0], dtype=torch.int32)
    L = _leaves(sc)
    scg = sc.to(device=DEV)
    opts = {"max_iter": 10, "compute": compute}
    v_new = _recorded_step("fused", sc, L, scg, count.to(DEV), scg.Je, opts)
    cot = torch.randn(sc.B, sc.nb, 3, generator=torch.Generator().manual_seed(12))
    (v_new * cot.to(DEV)).sum().backward()
    v_ref, g_ref = _oracle_step(sc, cot, count)
    tol_v, tol_g = ((2e-6, 1e-4) if pts == 2 else (1e-4, 1e-2)) if compute == "f64" else (5e-2, 2e-1)
    ev = _rel(v_new.detach().double().cpu(), v_ref)
    print("beyond 64 contacts:", compute, nbox, pts, "forward", ev.tolist())
    assert float(ev.max()) < tol_v
    for k in ("Mdiag", "v", "f"):
        err = _rel(L[k].grad.double().cpu(), g_ref[k])
        print("   ", k, err.tolist())
        assert float(err.max() if pts == 2 else err.median()) < tol_g, (k, err)
        assert bool(torch.isfinite(L[k].grad).all())
    for i, c in enumerate(count.tolist()):
        for k in ("c_n", "c_p1", "c_p2"):
            assert float(L[k].grad[i, c:].abs().max() if c < sc.nc else 0.0) == 0.0

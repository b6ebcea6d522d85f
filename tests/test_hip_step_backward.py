"""GPU parity of the fused analytic backward (`lcp_step_backward_f32`): d(loss)/d(v_new) -> gradients with respect to
the physical inputs of a step, against
  (a) the SAME device solution's dense LCP gradients (`lcp_pdipm_backward_f32`) contracted through the engine
      assembly by autograd of the oracle's restatement (engines.py:31-32,50-74; world.py:144-234) - every key, to
      fp32 rounding, because both sides use the same multipliers; and
  (b) the fp64 oracle end to end (oracle forward + `lcp.py:37-64` + autograd), within 1e-4 on the parameters whose
      gradients are well defined for the scene family (tests/parity.py::err_physical)."""
import pytest
import torch

from oracle import pdipm_oracle as O
from tests import parity

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(sc, cot_v):
    from lcp_physics_amd.lcp import lcp_backward
    from lcp_physics_amd.physics.batched_world import assemble_contacts, fused_step, fused_step_backward, solution_of_step
    scg = sc.to(device=DEV)
    out = fused_step(scg)
    pg = fused_step_backward(scg, out, cot_v.to(DEV), want_Je=True)
    lcp = assemble_contacts(scg)
    sol = solution_of_step(scg, out, lcp[2], lcp[4])
    dense = lcp_backward(sol, (-cot_v).reshape(sc.B, -1).to(DEV))            # d(loss)/dx = -d(loss)/d(v_new)
    torch.cuda.synchronize()
    dense = {k: (None if t is None else t.double().cpu()) for k, t in zip("QpGhAbF", dense)}
    return {k: v.double().cpu() for k, v in pg.items()}, dense, out


@pytest.mark.parametrize("nbox,pts", [(2, 2), (4, 2), (4, 4), (3, 4)])
def test_matches_autograd_contraction_of_the_dense_gradients(nbox, pts):
    from lcp_physics_amd import scenes
    B = 64
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=300 + 10 * nbox + pts, dtype=torch.float32)
    cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(8), dtype=torch.float32)
    pg, dense, out = _run(sc, cot)
    ph = {k: v.double() if v.is_floating_point() else v for k, v in sc.phys_dict().items()}
    ref = parity.physical_grads(ph, sc.dt, dense, O)
    for k in parity.PHYS_KEYS:
        scale = ref[k].abs().reshape(B, -1).max(dim=1)[0].clamp_min(1e-30)
        err = (pg[k] - ref[k]).abs().reshape(B, -1).max(dim=1)[0] / scale
        # the dense gradients were rounded to fp32 on their way out (dG, dF entries up to 1e4 larger than their sum)
        bound = 2e-3 if k in ("c_n", "c_p1", "c_p2", "rest", "fric") else 1e-4
        big = scale > 1e-6 * scale.max()
        assert float(err[big].max()) < bound, (k, float(err[big].max()), int(err.argmax()))
    # the joint Jacobian's gradient IS the dense dA (lcp.py:57) - no assembly in between
    assert sc.Je is not None and "Je" in pg
    scale = dense["A"].abs().reshape(B, -1).max(dim=1)[0].clamp_min(1e-30)
    err = (pg["Je"] - dense["A"]).abs().reshape(B, -1).max(dim=1)[0] / scale
    assert float(err.max()) < 1e-4, (float(err.max()), int(err.argmax()))


@pytest.mark.parametrize("nbox,pts", [(2, 2), (4, 4)])
def test_matches_oracle_end_to_end(nbox, pts):
    from lcp_physics_amd import scenes
    B = 64
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=77 + nbox, dtype=torch.float32)
    cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(9), dtype=torch.float32)
    pg, _, out = _run(sc, cot)
    lcp64 = [None if t is None else t.double() for t in O.assemble_lcp(*sc.assembly_args())]
    refsol = O.lcp_forward(*lcp64)
    cx = (-cot).reshape(B, -1).double()
    gref = O.lcp_backward(refsol, *lcp64, cx)
    gref = {k: gref["d" + k] for k in "QpGhAbF"}
    Q, p, G, h, A, b, F = lcp64
    res_o = parity.kkt_backward_residual(Q, G, A, F, refsol.z, refsol.s, cx, gref["p"], -gref["h"], -gref["b"])
    ok = torch.stack([v for v in res_o.values()]).max(dim=0)[0] < 1e-9
    zs, ss = refsol.z.max(dim=1, keepdim=True)[0], refsol.s.max(dim=1, keepdim=True)[0]
    ok = ok & (torch.maximum(refsol.z / zs, refsol.s / ss).min(dim=1)[0] > 1e-6)
    assert float(ok.float().mean()) >= parity.MIN_WELL_POSED_FRAC          # (oracle alone, these seeds: 100 % / 90.6 %)
    ph = {k: v.double() if v.is_floating_point() else v for k, v in sc.phys_dict().items()}
    pg_ref = parity.physical_grads(ph, sc.dt, gref, O)
    scl = parity.free_scales(Q, p, cx)
    floor = parity._n(cx) * torch.maximum(scl["x_free"], parity._n(refsol.x))
    ep = parity.err_physical(pg, pg_ref, ph, floor, keys=["Mdiag", "v", "f"])     # (redundant rows: see err_physical)
    assert float(ep[ok].max()) < 1e-4, (float(ep[ok].max()), int(ep.argmax()))


def test_variable_contact_counts_and_padding():
    """After lcp_solve_dynamics_f32 with per-scene counts: padded slots get zero geometry gradients and a scene
    without contacts gets the free-body gradient d v = M^-1 ... (x = -M^-1 u: dMdiag, dv, df only)."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics.batched_world import fused_step_backward, solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    B = 8
    sc = scenes.make_stack_scenes(B=B, nbox=2, pts_per_interface=4, seed=5, dtype=torch.float32).to(device=DEV)
    cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2
    count = torch.tensor([8, 4, 0, 8, 2, 0, 6, 8], dtype=torch.int32, device=DEV)
    out = solve_dynamics(B, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.f, sc.rest, sc.fric, cb, sc.Je, sc.dt)
    cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(1), dtype=torch.float32).to(DEV)
    pg = fused_step_backward(sc, out, cot)
    torch.cuda.synchronize()
    slot = torch.arange(sc.nc, device=DEV).unsqueeze(0) >= count.unsqueeze(1)
    for k in ("c_n", "c_p1", "c_p2"):
        assert float(pg[k][slot].abs().max()) == 0.0
        assert bool(torch.isfinite(pg[k]).all())
    # no contact, floor pinned by the joint: boxes are free bodies, v_new = v + dt f / M
    free = (count == 0)
    dv_expected = cot[free][:, 1:]                                  # d v_new / d v = 1 for the free boxes
    assert torch.allclose(pg["v"][free][:, 1:], dv_expected, rtol=1e-5, atol=1e-6)
    assert torch.allclose(pg["f"][free][:, 1:], cot[free][:, 1:] * sc.dt / sc.Mdiag[free][:, 1:], rtol=1e-5, atol=1e-7)
    assert float(pg["rest"][free].abs().max()) == 0.0 and float(pg["fric"][free].abs().max()) == 0.0


@pytest.mark.parametrize("nbox,pts", [(5, 2), (8, 2), (9, 1), (6, 4), (11, 1)])
def test_mid_size_backward_matches_generic_dense(nbox, pts):
    """6 / 9 / 10 bodies with 10 / 16 / 9 contacts (the nz <= 32 instantiation of lcp_quad.hip) and 7 bodies / 24
    contacts (the 32-contact class of lcp_big.hip): backward against the generic kernels' dense gradients contracted by
    autograd (parameters entering through Q and p)."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_backward
    from lcp_physics_amd.physics.batched_world import assemble_contacts, fused_step, fused_step_backward, solution_of_step, solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    B = 16
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=60 + nbox, dtype=torch.float32)
    scg = sc.to(device=DEV)
    cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(13), dtype=torch.float32)
    cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    count = torch.full((B,), sc.nc, dtype=torch.int32, device=DEV)
    out = solve_dynamics(B, sc.nb, sc.nc, 3, count, scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, scg.Je, sc.dt)
    pg = {k: v.double().cpu() for k, v in fused_step_backward(scg, out, cot.to(DEV), want_Je=True).items()}
    gen = fused_step(scg, path="generic")
    lcp = assemble_contacts(scg)
    dense = lcp_backward(solution_of_step(scg, gen, lcp[2], lcp[4]), (-cot).reshape(B, -1).to(DEV))
    torch.cuda.synchronize()
    dense = {k: (None if t is None else t.double().cpu()) for k, t in zip("QpGhAbF", dense)}
    ph = {k: v.double() if v.is_floating_point() else v for k, v in sc.phys_dict().items()}
    ref = parity.physical_grads(ph, sc.dt, dense, O)
    for k in ("Mdiag", "v", "f"):
        scale = ref[k].abs().reshape(B, -1).max(dim=1)[0].clamp_min(1e-30)
        err = (pg[k] - ref[k]).abs().reshape(B, -1).max(dim=1)[0] / scale
        big = scale > 1e-6 * scale.max()
        assert float(err[big].max()) < 2e-3, (k, float(err[big].max()), int(err.argmax()))
    scale = dense["A"].abs().reshape(B, -1).max(dim=1)[0].clamp_min(1e-30)                # dJe = the dense dA (lcp.py:57)
    err = (pg["Je"] - dense["A"]).abs().reshape(B, -1).max(dim=1)[0] / scale
    assert float(err.max()) < 2e-3, ("Je", float(err.max()), int(err.argmax()))


def test_large_scene_backward_matches_generic_dense_and_oracle():
    """Config-5 sized piles (11 bodies, 64 contacts): `lcp_step_backward_f32` after `lcp_solve_dynamics_f32` (both in
    lcp_big.hip) against (a) the generic kernels' dense backward contracted by autograd and (b) the fp64 oracle."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_backward
    from lcp_physics_amd.physics.batched_world import assemble_contacts, fused_step, fused_step_backward, solution_of_step, solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    B = 8
    sc = scenes.make_pile_scenes(B=B, seed=31, dtype=torch.float32)
    scg = sc.to(device=DEV)
    cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(3), dtype=torch.float32)
    cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    count = torch.full((B,), sc.nc, dtype=torch.int32, device=DEV)
    # (pinned=True: the LCP_HINT_PINNED that fused_step derives from the scene's Je itself - the same word, the same kernels)
    out = solve_dynamics(B, sc.nb, sc.nc, 3, count, scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, scg.Je, sc.dt, pinned=True)
    pg = {k: v.double().cpu() for k, v in fused_step_backward(scg, out, cot.to(DEV)).items()}
    # either forward entry picks the same kernel family for these sizes: same workspace, same gradients
    pg2 = fused_step_backward(scg, fused_step(scg), cot.to(DEV))
    for k in pg:
        assert torch.equal(pg2[k].double().cpu(), pg[k]), k
    # (round 6) the generic kernels keep their iterate as well: lcp_step_bwd_kernel on the same scenes, the same gradients
    pg3 = fused_step_backward(scg, fused_step(scg, path="generic"), cot.to(DEV))
    for k in ("Mdiag", "v", "f"):
        scale = pg[k].abs().reshape(B, -1).max(dim=1)[0].clamp_min(1e-30)
        assert float(((pg3[k].double().cpu() - pg[k]).abs().reshape(B, -1).max(dim=1)[0] / scale).max()) < 1e-3, k
    # (a) generic kernels: dense gradients of the same step, contracted through the assembly by autograd
    gen = fused_step(scg, path="generic")
    lcp = assemble_contacts(scg)
    dense = lcp_backward(solution_of_step(scg, gen, lcp[2], lcp[4]), (-cot).reshape(B, -1).to(DEV))
    torch.cuda.synchronize()
    dense = {k: (None if t is None else t.double().cpu()) for k, t in zip("QpGhAbF", dense)}
    ph = {k: v.double() if v.is_floating_point() else v for k, v in sc.phys_dict().items()}
    ref = parity.physical_grads(ph, sc.dt, dense, O)
    for k in ("Mdiag", "v", "f"):                                                # defined whatever the multipliers (err_physical)
        scale = ref[k].abs().reshape(B, -1).max(dim=1)[0].clamp_min(1e-30)
        err = (pg[k] - ref[k]).abs().reshape(B, -1).max(dim=1)[0] / scale
        assert float(err.max()) < 1e-3, (k, float(err.max()))
    # (b) the oracle end to end
    lcp64 = [None if t is None else t.double() for t in O.assemble_lcp(*sc.assembly_args())]
    refsol = O.lcp_forward(*lcp64)
    cx = (-cot).reshape(B, -1).double()
    gref = O.lcp_backward(refsol, *lcp64, cx)
    gref = {k: gref["d" + k] for k in "QpGhAbF"}
    Q, p, G, h, A, b, F = lcp64
    res_o = parity.kkt_backward_residual(Q, G, A, F, refsol.z, refsol.s, cx, gref["p"], -gref["h"], -gref["b"])
    ok = torch.stack([v for v in res_o.values()]).max(dim=0)[0] < 1e-9
    zs, ss = refsol.z.max(dim=1, keepdim=True)[0], refsol.s.max(dim=1, keepdim=True)[0]
    ok = ok & (torch.maximum(refsol.z / zs, refsol.s / ss).min(dim=1)[0] > 1e-6)
    pg_ref = parity.physical_grads(ph, sc.dt, gref, O)
    scl = parity.free_scales(Q, p, cx)
    floor = parity._n(cx) * torch.maximum(scl["x_free"], parity._n(refsol.x))
    ep = parity.err_physical(pg, pg_ref, ph, floor, keys=["Mdiag", "v", "f"])
    print("well-posed scenes", int(ok.sum()), "of", B)
    assert int(ok.sum()) >= B // 2, ("too few scenes whose backward system the oracle itself solves", int(ok.sum()), B)
    assert float(ep[ok].max()) < 1e-4, (float(ep[ok].max()), int(ok.sum()))


def test_batched_world_differentiable_step_is_the_same_backward_as_a_graph_node():
    """`BatchedWorld.step(differentiable=True)`: torch autograd through `SolveDynamicsFunction` and `p + v dt` returns exactly
    what `fused_step_backward` returns for the same cotangent, and two chained steps back-propagate through both."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics.batched_world import BatchedWorld, fused_step, fused_step_backward
    B = 32
    sc = scenes.make_stack_scenes(B=B, nbox=3, pts_per_interface=2, seed=91, dtype=torch.float32).to(device=DEV)
    cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(4), dtype=torch.float32).to(DEV)
    out = fused_step(sc)
    ref = fused_step_backward(sc, out, cot)
    leaves = {}
    for k in ("Mdiag", "v", "f", "rest", "fric"):
        leaves[k] = getattr(sc, k).clone().requires_grad_(True)
    from dataclasses import replace
    world = BatchedWorld(replace(sc, **leaves))
    r = world.step(differentiable=True)
    assert float((r["v_new"].detach() - out["v_new"]).abs().max()) == 0.0
    (r["v_new"] * cot).sum().backward()
    for k in leaves:
        assert float((leaves[k].grad - ref[k]).abs().max()) <= 1e-6 * max(1.0, float(ref[k].abs().max())), k
    # two steps: the gradient of the second step's velocities reaches the first step's inputs
    for t in leaves.values():
        t.grad = None
    world = BatchedWorld(replace(sc, **leaves))
    world.step(differentiable=True)
    r2 = world.step(differentiable=True)
    (r2["p_new"] * cot).sum().backward()
    assert all(t.grad is not None and bool(torch.isfinite(t.grad).all()) for t in leaves.values())
    assert float(leaves["f"].grad.abs().max()) > 0


def test_dense_backward_follows_either_forward_entry_where_the_two_pick_different_kernels():
    """fp32 arithmetic, nz <= 16 with 5..8 equality rows: `lcp_step_fused_f32` (full lists) runs the wave64 step kernel,
    `lcp_solve_dynamics_f32` (a contact count per scene) the generic one - the `compute` word cannot tell the two apart, the tag in
    the workspace trailer can.  `lcp_pdipm_backward_f32` + LCP_HINT_ALL_CONTACT after EITHER forward returns that forward's
    gradients (it used to return NaN after the second); plan-cache hits must follow replaced output tensors."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_backward
    from lcp_physics_amd.physics.batched_world import assemble_contacts, fused_step, solution_of_step, solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    B = 24
    sc = scenes.make_stack_scenes(B=B, nbox=3, pts_per_interface=4, seed=61, dtype=torch.float32)
    nz = 3 * sc.nb
    Je = torch.zeros(B, 6, nz)
    Je[:, :3, :3] = torch.eye(3)
    Je[:, 3, 3], Je[:, 3, 6] = 1.0, -1.0                         # boxes 1 and 2 turn together
    Je[:, 4, 6], Je[:, 4, 9] = 1.0, -1.0                         # boxes 2 and 3 turn together
    Je[:, 5, 4], Je[:, 5, 7] = 1.0, -1.0                         # boxes 1 and 2 slide together
    sc.Je = Je
    scg = sc.to(device=DEV)
    lcp = assemble_contacts(scg)
    cot = torch.randn(B, nz, generator=torch.Generator().manual_seed(2), dtype=torch.float32).to(DEV)
    a = fused_step(scg, compute="f32")
    ga = lcp_backward(solution_of_step(scg, a, lcp[2], lcp[4], compute="f32"), cot)
    cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    count = torch.full((B,), sc.nc, dtype=torch.int32, device=DEV)
    b = solve_dynamics(B, sc.nb, sc.nc, 6, count, scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, scg.Je, sc.dt, compute="f32")
    sb = solution_of_step(scg, b, lcp[2], lcp[4], compute="f32")
    gb = lcp_backward(sb, cot)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(a["v_new"]).all()) and bool(torch.isfinite(b["v_new"]).all())
    vs = float(a["v_new"].abs().max())
    assert float((a["v_new"] - b["v_new"]).abs().max()) <= 2e-3 * vs          # (all-fp32 arithmetic, two elimination orders)
    for k, x, y in zip("QpGhAbF", ga, gb):
        assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(y).all()), k
    dps = float(ga[1].abs().max())
    assert float((ga[1] - gb[1]).abs().max()) <= 2e-2 * dps
    # the cached argument list of the backward follows a replaced output tensor (it used to keep writing the old one)
    keep = gb[1]
    gb2 = list(gb)
    gb2[1] = torch.zeros_like(keep)
    sb._bwd_plan = (gb2,) + tuple(sb._bwd_plan[1:])                           # (the handle the first call cached, now with a new dp tensor)
    lcp_backward(sb, cot, out=gb2)
    torch.cuda.synchronize()
    assert torch.equal(gb2[1], keep) and gb2[1].data_ptr() != keep.data_ptr()
    # ... and so does the step's: replace v_new in the handle, the next call must fill the new tensor
    old_v = a["v_new"]
    a["v_new"] = torch.zeros_like(old_v)
    a = fused_step(scg, compute="f32", ws=a["ws"], out=a)
    torch.cuda.synchronize()
    assert torch.equal(a["v_new"], old_v) and a["v_new"].data_ptr() != old_v.data_ptr()

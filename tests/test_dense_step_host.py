"""CPU: the differentiable torch assembly behind the dense-boundary fallback of the contact-list entry points
(`lcp_physics_amd/physics/dense_step.py`; reference `engines.py:31-32,50-74,80-116`, `world.py:144-234`) against the oracle's
restatement, values and autograd; the linear solve of scenes without contacts (`engines.py:36-49`) against the explicit inverse."""
import torch

from lcp_physics_amd import scenes
from lcp_physics_amd.physics import dense_step as D
from oracle import pdipm_oracle as O


def _scene(B=3, nbox=4, pts=2, seed=5):
    return scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=seed, dtype=torch.float64)


def test_dynamics_assembly_equals_the_oracle():
    sc = _scene()
    ref = O.assemble_lcp(*sc.assembly_args())
    got = D.assemble_dynamics(sc.Mdiag, sc.v, sc.f, sc.rest, sc.fric, sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2, sc.Je, sc.dt)
    for name, a, b in zip("QpGhAbF", got, ref):
        assert torch.equal(a, b), name


def test_post_stabilization_assembly_equals_the_oracle():
    sc = _scene(nbox=3, pts=4)
    ref = O.assemble_post_stabilization(sc.Mdiag, sc.v, sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2, sc.rest, sc.Je)
    got = D.assemble_post_stabilization(sc.Mdiag, sc.v, sc.rest, sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2, sc.Je)
    for name, a, b in zip("QpGhAbF", got, ref):
        assert torch.equal(a, b), name


def test_assembly_autograd_equals_the_oracles():
    sc = _scene()
    keys = ("Mdiag", "v", "f", "rest", "fric", "c_n", "c_p1", "c_p2")

    def grads(assemble):
        leaves = {k: getattr(sc, k).clone().requires_grad_(True) for k in keys}
        lcp = assemble(leaves)
        g = torch.Generator().manual_seed(1)
        loss = sum((t * torch.randn(t.shape, generator=g, dtype=t.dtype)).sum() for t in lcp if t is not None and t.requires_grad)
        return torch.autograd.grad(loss, [leaves[k] for k in keys])

    mine = grads(lambda L: D.assemble_dynamics(L["Mdiag"], L["v"], L["f"], L["rest"], L["fric"], L["c_n"], L["c_p1"], L["c_p2"],
                                               sc.c_i1, sc.c_i2, sc.Je, sc.dt))
    ref = grads(lambda L: O.assemble_lcp(L["Mdiag"], L["v"], L["f"], sc.dt, L["c_n"], L["c_p1"], L["c_p2"], sc.c_i1, sc.c_i2,
                                         L["rest"], L["fric"], sc.Je))
    for k, a, b in zip(keys, mine, ref):
        assert torch.allclose(a, b, rtol=1e-13, atol=1e-13), k


def test_scenes_without_contacts_take_the_linear_solve_of_the_reference():
    B, nz, e = 4, 9, 3
    g = torch.Generator().manual_seed(2)
    Md = torch.rand(B, nz, generator=g, dtype=torch.float64) + 0.5
    Je = torch.randn(B, e, nz, generator=g, dtype=torch.float64)
    top, bottom = torch.randn(B, nz, generator=g, dtype=torch.float64), torch.randn(B, e, generator=g, dtype=torch.float64)
    # engines.py:38-49: P = [[M, -Je^T], [Je, 0]], x = inverse(P) u
    P = torch.cat([torch.cat([torch.diag_embed(Md), -Je.transpose(1, 2)], dim=2),
                   torch.cat([Je, torch.zeros(B, e, e, dtype=torch.float64)], dim=2)], dim=1)
    x = (torch.inverse(P) @ torch.cat([top, bottom], dim=1).unsqueeze(2)).squeeze(2)[:, :nz]
    assert torch.allclose(D._linear(Md, top, Je, bottom), x, rtol=1e-10, atol=1e-12)
    assert torch.equal(D._linear(Md, top, None, None), top / Md)


def test_scenes_are_grouped_by_their_contact_count():
    count = torch.tensor([3, 0, 5, 3, 9], dtype=torch.int32)
    groups, truncated = D._groups(count, 5, 5, "cpu")
    assert [(c, i.tolist()) for c, i in groups] == [(0, [1]), (3, [0, 3]), (5, [2, 4])]
    assert truncated.tolist() == [False, False, False, False, True]
    groups, truncated = D._groups(torch.full((4,), 5, dtype=torch.int32), 5, 4, "cpu")
    assert groups == [(5, None)] and not bool(truncated.any())
    assert D._groups(None, 7, 2, "cpu") == ([(7, None)], None)

"""GPU: lcp_solo.hip - the contact-list step with ONE scene per wavefront (small batches) - against the four-scenes-per-wave kernel
it stands in for (same equations, other summation orders: agreement to rounding), against the fp64 oracle, with ragged contact
counts, with equality rows it has to hand on to the general kernel, and under both backward kernels."""
import pytest
import torch

from oracle import pdipm_oracle as O
from tests import parity

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _both(sc, **kw):
    from lcp_physics_amd.physics import fused_step
    a = fused_step(sc, path="solo", **kw)
    b = fused_step(sc, path="quad", **kw)
    torch.cuda.synchronize()
    return a, b


@pytest.mark.parametrize("nbox,pts,B", [(2, 4, 97), (4, 4, 64), (4, 2, 33), (3, 4, 1), (1, 4, 5)])
def test_solo_equals_quad_and_oracle(nbox, pts, B):
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics import assemble_contacts
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=100 + nbox, dtype=torch.float32)
    scg = sc.to(device=DEV)
    a, b = _both(scg)
    va, vb = a["v_new"].double().cpu(), b["v_new"].double().cpu()
    scale = vb.abs().reshape(B, -1).max(dim=1)[0].clamp_min(1.0)
    assert float(((va - vb).abs().reshape(B, -1).max(dim=1)[0] / scale).max()) <= 1e-6
    assert torch.allclose(a["p_new"], b["p_new"], rtol=1e-5, atol=1e-4)
    assert int((a["status"] & ~4).max()) == 0 and int((a["iters"] - b["iters"]).abs().max()) <= 2
    lcp = [None if t is None else t.double().cpu() for t in assemble_contacts(scg)]
    rep, ref = parity.headline_report(O, lcp, -va.reshape(B, -1), a["z"].cpu(), a["s"].cpu(), a["iters"].cpu())
    assert rep["fwd_err_x_max"] <= 1e-4, rep
    assert rep["index_set_mismatches_floor_0.0001"] == 0, rep
    assert rep["iters_max_abs_delta"] <= 2, rep
    ey = (a["y"].double().cpu() - ref.y).abs().max(dim=1)[0] / ref.y.abs().max(dim=1)[0].clamp_min(1.0)
    assert float(ey.max()) <= 1e-4


def test_solo_ragged_contact_counts_and_no_contact_scenes():
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics.batched_world import solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    B = 40
    sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=9, dtype=torch.float32)
    scg = sc.to(device=DEV)
    cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    count = torch.randint(0, sc.nc + 1, (B,), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    count[::5] = 0
    count[1::7] = sc.nc + 3                                    # overflow: solved with the first nc contacts, flagged
    run = lambda path: solve_dynamics(B, sc.nb, sc.nc, 3, count.to(DEV), scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, scg.Je, sc.dt, path=path)
    a, b = run("solo"), run("quad")
    torch.cuda.synchronize()
    va, vb = a["v_new"].double().cpu(), b["v_new"].double().cpu()
    scale = vb.abs().reshape(B, -1).max(dim=1)[0].clamp_min(1.0)
    assert float(((va - vb).abs().reshape(B, -1).max(dim=1)[0] / scale).max()) <= 1e-6
    assert torch.equal(a["status"] & 16, b["status"] & 16) and int((a["status"][1::7] & 16).min()) == 16
    assert float((a["z"] - b["z"]).abs().max()) <= 1e-3 * float(b["z"].abs().max())
    # padded slots report zero multipliers
    slot = torch.arange(sc.nc).unsqueeze(0) >= count.clamp(max=sc.nc).unsqueeze(1)
    assert float(a["z"].cpu()[:, :sc.nc][slot].abs().max()) == 0.0


def test_solo_hands_general_equality_rows_to_the_general_kernel():
    from lcp_physics_amd import scenes
    B = 24
    sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=17, dtype=torch.float32)
    Je = sc.Je.clone()
    Je[1::3] *= 2.0                                            # the same constraint, not the pinned form
    Je[2::3, 1, 3] = 0.25                                      # a row with a general entry
    sc.Je = Je
    a, b = _both(sc.to(device=DEV))
    assert torch.allclose(a["v_new"], b["v_new"], rtol=0, atol=1e-5 * float(b["v_new"].abs().max()))
    assert torch.allclose(a["y"], b["y"], rtol=1e-4, atol=1e-4 * float(b["y"].abs().max()))
    assert int((a["status"] & ~4).max()) == 0


def test_backward_kernels_follow_a_solo_forward():
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import lcp_backward
    from lcp_physics_amd.physics import assemble_contacts
    from lcp_physics_amd.physics.batched_world import fused_step_backward, solution_of_step
    B = 48
    scg = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=23, dtype=torch.float32).to(device=DEV)
    lcp = assemble_contacts(scg)
    a, b = _both(scg)
    cot = torch.randn(B, 3 * scg.nb, generator=torch.Generator().manual_seed(1)).to(DEV)
    ga = lcp_backward(solution_of_step(scg, a, lcp[2], lcp[4]), cot)
    gb = lcp_backward(solution_of_step(scg, b, lcp[2], lcp[4]), cot)
    pa = fused_step_backward(scg, a, cot.reshape(B, scg.nb, 3).contiguous())
    pb = fused_step_backward(scg, b, cot.reshape(B, scg.nb, 3).contiguous())
    torch.cuda.synchronize()
    for k in (0, 1):                                           # dQ, dp: defined whatever the multipliers (tests/parity.py)
        assert bool(torch.isfinite(ga[k]).all())
        assert float((ga[k] - gb[k]).abs().max()) <= 1e-4 * max(float(gb[k].abs().max()), 1e-12)
    for k in ("Mdiag", "v", "f"):
        assert float((pa[k] - pb[k]).abs().max()) <= 1e-4 * max(float(pb[k].abs().max()), 1e-12), k


def test_pinned_hint_skips_the_general_pass_and_a_broken_promise_is_loud():
    """LCP_HINT_PINNED: the caller asserts Je = [I 0] (checked once per SceneBatch on the host by the wrappers) - same results,
    one launch less; a scene that breaks an explicit promise gets NaN velocities and LCP_ST_NAN, not a silently stale output."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics import fused_step
    from lcp_physics_amd.physics.batched_world import rows_pin_leading_coordinates
    B = 16
    sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=5, dtype=torch.float32)
    assert rows_pin_leading_coordinates(sc.Je)
    scg = sc.to(device=DEV)
    for path in ("quad", "solo"):
        a = fused_step(scg, path=path, pinned=True)
        b = fused_step(scg, path=path, pinned=False)
        torch.cuda.synchronize()
        assert torch.equal(a["v_new"], b["v_new"]) and torch.equal(a["z"], b["z"]) and torch.equal(a["iters"], b["iters"])
    Je = sc.Je.clone()
    Je[5] *= 2.0                                               # scene 5 (wave 1 of the four-scenes kernel) is not in the pinned form
    sc2 = scenes.SceneBatch(**{**{f: getattr(sc, f) for f in ("p", "v", "Mdiag", "f", "rest", "fric", "c_n", "c_p1", "c_p2", "c_i1", "c_i2")},
                               "Je": Je, "dt": sc.dt}).to(device=DEV)
    assert not rows_pin_leading_coordinates(sc2.Je)
    ok = fused_step(sc2)                                        # the default: detected, the general pass runs
    bad = fused_step(sc2, path="solo", pinned=True)             # an explicit (wrong) promise
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ok["v_new"]).all())
    assert bool(torch.isnan(bad["v_new"][5]).all()) and int(bad["status"][5]) & 8
    assert bool(torch.isfinite(bad["v_new"][:5]).all()) and bool(torch.isfinite(bad["v_new"][6:]).all())

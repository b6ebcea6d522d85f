"""CPU: the comparison helpers of tests/parity.py checked on the oracle itself - a gate that cannot fail is no gate.

`own_iterate_backward` (round 5): the kernel's backward against the oracle's backward evaluated at the kernel's own iterate.  Fed with the
oracle's own gradients it must report zero error on the scenes it calls determined; fed with a gradient that solves the system only up to
a null-space blow-up (what the contact-space kernel returned on one converged scene before round 5: multipliers of 1e15, dx off by per
cent, residual 1e-16 of their size) it must report the error although `kkt_backward_residual` - the gate round 5 started with - does not."""
import copy

import torch

from lcp_physics_amd import scenes
from oracle import pdipm_oracle as O
from tests import parity


def _case(B=24, seed=4242):
    sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=seed, dtype=torch.float64)
    lcp = [None if t is None else t.double() for t in O.assemble_lcp(*sc.assembly_args())]
    ref = O.lcp_forward(*lcp)
    cot = torch.randn(B, lcp[0].shape[1], generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    g = O.lcp_backward(ref, *lcp, cot)
    fl = parity.grad_floors(lcp[0], lcp[1], cot, ref.x, ref.z, ref.y)
    return lcp, ref, cot, g, fl


def test_own_iterate_gate_is_silent_on_the_oracle_itself():
    lcp, ref, cot, g, fl = _case()
    rep = parity.own_iterate_backward(O, lcp, ref, cot, ref.x, ref.z, ref.s, {k: g["d" + k] for k in "pQb"}, fl)
    assert rep["bwd_own_iterate_determined_scenes"] >= 16, rep
    assert rep["bwd_own_iterate_err_max"] <= 1e-12, rep


def test_own_iterate_gate_sees_what_the_residual_gate_cannot():
    lcp, ref, cot, g, fl = _case()
    Q, p, G, h, A, b, F = lcp
    B, m, nz = G.shape
    nc = m // 4
    # a direction the converged system does not see: on a STICKING contact (cone row inactive: z_g << s_g; both friction multipliers
    # active: s_f << z_f) dlam_f1 = dlam_f2 = t leaves G^T dlam alone (world.py:191-192: rows +jt, -jt), the cone row takes it with
    # dlam_g = 2 t z_g / s_g, and the friction rows see (s_f / z_f) t - nothing, relative to t.  A huge t hides a wrong dx behind
    # the block norms of the residual.
    z, s_ = ref.z, ref.s
    c = torch.arange(nc)
    f1, f2, gm = nc + 2 * c, nc + 2 * c + 1, 3 * nc + c
    stick = (z[:, gm] / s_[:, gm] < 1e-8) & (s_[:, f1] / z[:, f1] < 1e-8) & (s_[:, f2] / z[:, f2] < 1e-8)
    has = stick.any(dim=1)
    assert int(has.sum()) >= 8, int(has.sum())
    t = 1e12 * stick.double()
    bad = {k: g["d" + k].clone() for k in "pQb"}
    dlam = -g["dh"].clone()
    dlam[:, f1] += t
    dlam[:, f2] += t
    dlam[:, gm] += 2 * t * z[:, gm] / s_[:, gm]
    free = torch.ones(nz, dtype=torch.float64)
    free[:3] = 0                                                                  # (A = [I 0] pins the floor: A dx = 0 stays true)
    bad["p"] = bad["p"] + has.double().unsqueeze(1) * 0.05 * parity._n(g["dp"]).unsqueeze(1) * free / nz ** 0.5
    res = parity.kkt_backward_residual(Q, G, A, F, ref.z, ref.s, cot, bad["p"], dlam, -g["db"])
    assert float(torch.stack(list(res.values()))[:, has].max()) < 1e-6           # the residual gate passes it ...
    rep = parity.own_iterate_backward(O, lcp, ref, cot, ref.x, ref.z, ref.s, bad, fl)
    assert rep["bwd_own_iterate_err_max"] > 1e-3, rep                            # ... the own-iterate gate does not


def test_own_iterate_gate_does_not_count_a_scene_the_oracle_cannot_determine():
    lcp, ref, cot, g, fl = _case()
    # make one scene's system singular to working precision: a contact pair with z_i ~ s_i ~ 1e-17 (what an over-converged solve leaves)
    at = copy.copy(ref)
    at.z, at.s = ref.z.clone(), ref.s.clone()
    at.z[0, :4], at.s[0, :4] = 1e-17, 1e-17
    full = parity.own_iterate_backward(O, lcp, ref, cot, ref.x, ref.z, ref.s, {k: g["d" + k] for k in "pQb"}, fl)
    g_at = O.lcp_backward(at, *lcp, cot)
    rep = parity.own_iterate_backward(O, lcp, ref, cot, ref.x, at.z, at.s, {k: g_at["d" + k] for k in "pQb"}, fl)
    assert rep["bwd_own_iterate_determined_scenes"] <= full["bwd_own_iterate_determined_scenes"], (rep, full)


def test_headline_report_on_the_oracle_itself_has_every_gated_field_and_no_error():
    """`headline_report` fed with the oracle's own iterate and gradients (what a perfect kernel would return): every field the GPU gates
    read (tests/test_hip_headline_parity.py::check_gates) exists and sits at its ideal value - the report's code paths run on CPU."""
    lcp, ref, cot, g, fl = _case(B=16)
    sc = scenes.make_stack_scenes(B=16, nbox=4, pts_per_interface=4, seed=4242, dtype=torch.float64)
    grads = {k: g["d" + k] for k in "QpGhAbF"}
    rep, _ = parity.headline_report(O, lcp, ref.x, ref.z, ref.s, ref.iters, cot=cot, grads=grads, phys=sc.phys_dict(), dt=sc.dt,
                                    input_stability=True, all_grads=True)
    for k in ("fwd_err_x_max", "index_set_mismatches_unmasked", "index_set_mismatches_floor_0.0001", "index_set_masked_frac", "iters_max_abs_delta",
              "bwd_nonfinite_scenes", "bwd_kkt_resid_all_max", "bwd_own_iterate_determined_scenes", "bwd_own_iterate_err_max",
              "bwd_well_posed_frac", "bwd_err_dp_max", "bwd_err_phys_max", "bwd_err_dQ_max", "bwd_err_dA_max", "bwd_err_db_max", "bwd_kkt_resid_max",
              "bwd_err_phys_all_max", "bwd_err_phys_five_max", "bwd_err_dlam_determined_max"):
        assert k in rep, k
    assert rep["fwd_err_x_max"] == 0.0 and rep["index_set_mismatches_unmasked"] == 0 and rep["iters_max_abs_delta"] == 0
    assert rep["bwd_nonfinite_scenes"] == 0
    assert rep["bwd_err_dp_max"] == 0.0 and rep["bwd_err_phys_max"] <= 1e-12 and rep["bwd_own_iterate_err_max"] <= 1e-12, rep


def test_index_set_accounting_explains_iterate_ties_and_catches_a_real_flip():
    """Round 6 (`index_set_mismatches_oracle_stable`): a "kernel" that returns the oracle's own PREVIOUS iterate on a few converged
    scenes differs from the oracle only on rows the oracle itself does not decide reproducibly (fp32 arithmetic, no pivoting, another
    order of the unknowns, or another iterate of its own trajectory): zero unexplained rows.  A planted flip of the most decisive row of
    one scene is none of those: it must be counted, and listed."""
    sc = scenes.make_stack_scenes(B=128, nbox=2, pts_per_interface=4, seed=1236, dtype=torch.float32)
    lcp = [None if t is None else t.double() for t in O.assemble_lcp(*sc.assembly_args())]
    tr = []
    ref = O.lcp_forward(*lcp, trace=tr)
    z, s = ref.z.clone(), ref.s.clone()
    xf = parity.free_scales(lcp[0], lcp[1])["x_free"]
    cand = [i for i in range(1, 128) if bool(((tr[-2]["z"][i] > tr[-2]["s"][i]) != (ref.z[i] > ref.s[i])).any()) and torch.equal(tr[-1]["z"][i], ref.z[i])
            and float((tr[-2]["x"][i] - ref.x[i]).norm() / xf[i]) <= 1e-8]                       # (converged solves: the last two iterates are the same solution)
    assert len(cand) >= 3
    for i in cand[:3]:
        z[i], s[i] = tr[-2]["z"][i], tr[-2]["s"][i]
    rep, _ = parity.headline_report(O, lcp, ref.x, z, s, ref.iters)
    assert rep["index_set_mismatches_unmasked"] > 0 and rep["index_set_mismatches_oracle_stable"] == 0, rep
    assert rep["ref_fp32_vs_fp64_index_rows"] > 0 and rep["ref_fp32_vs_fp64_err_x_max"] > 0, rep      # (SURVEY 8d's companion fields ride along)
    big = int((ref.z[0] - ref.s[0]).abs().argmax())
    z[0, big], s[0, big] = ref.s[0, big].clone(), ref.z[0, big].clone()
    rep, _ = parity.headline_report(O, lcp, ref.x, z, s, ref.iters)
    assert rep["index_set_mismatches_oracle_stable"] == 1, rep
    bad = [r for r in rep["index_set_mismatch_rows"] if not (r["oracle_flips_in_fp32"] or r["oracle_flips_without_pivoting"] or r["oracle_flips_reordered"]
                                                              or r["kernel_set_is_another_oracle_iterate"])]
    assert [(r["scene"], r["row"]) for r in bad] == [(0, big)], bad

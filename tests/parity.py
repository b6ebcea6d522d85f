"""Parity metrics shared by the CPU (oracle-vs-reference) and GPU (HIP-vs-oracle) tests.

SURVEY.md §8(d): a resting stack has ||x_ref|| ~ 1e-15 and analytically-zero gradients,
so plain relative errors are meaningless there.  Errors are therefore scaled by the
free-motion magnitudes (what the quantity would be with no contact constraint):

    x_free  = Q^-1 p            dx_free = Q^-1 dl_dx
    err_x   = ||x - x_ref|| / max(||x_ref||, ||x_free||)
    err_g   = ||g - g_ref||_F / max(||g_ref||_F, S_g)

with dimensionally consistent floors S_g built from ||dx_free||, ||x_free||, ||z_ref||,
||y_ref|| (see `grad_floors`).  All norms are per scene.
"""
import torch


def _n(t):
    return t.reshape(t.shape[0], -1).double().norm(dim=1)


def free_scales(Q, p, cot=None):
    Qd = Q.double()
    xf = torch.linalg.solve(Qd, p.double().unsqueeze(-1)).squeeze(-1)
    out = {"x_free": _n(xf)}
    if cot is not None:
        df = torch.linalg.solve(Qd, cot.double().unsqueeze(-1)).squeeze(-1)
        out["dx_free"] = _n(df)
    return out


def err_x(x, x_ref, Q, p):
    sc = free_scales(Q, p)["x_free"]
    return _n(x.double() - x_ref.double()) / torch.maximum(_n(x_ref), sc).clamp_min(1e-300)


def rel_err(a, a_ref, floor=1e-300):
    return _n(a.double() - a_ref.double()) / _n(a_ref).clamp_min(floor)


def grad_floors(Q, p, cot, x_ref, z_ref, y_ref=None):
    sc = free_scales(Q, p, cot)
    xf = torch.maximum(sc["x_free"], _n(x_ref)).clamp_min(1e-300)
    df = sc["dx_free"]
    zn = _n(z_ref)
    fl = {"p": df, "Q": df * xf, "G": df * zn, "h": df * zn / xf, "F": df * zn * zn / xf}
    if y_ref is not None:
        yn = _n(y_ref)
        fl["A"] = df * torch.maximum(yn, zn)
        fl["b"] = df * torch.maximum(yn, zn) / xf
    return fl


def err_grads(grads, grads_ref, floors):
    """grads / grads_ref: dicts keyed by 'Q','p','G','h','A','b','F' (None entries skipped)."""
    out = {}
    for k, g_ref in grads_ref.items():
        if g_ref is None or grads.get(k) is None:
            continue
        den = torch.maximum(_n(g_ref), floors[k]).clamp_min(1e-300)
        out[k] = _n(grads[k].double() - g_ref.double()) / den
    return out


def active_sets(z, s, nc=None):
    """Contact index set {i : z_i > s_i}.  Returns a bool mask [B,m]."""
    return z > s


def decisive_rows(z, s, rel=1e-3, floor=1e-5):
    """Rows where the ORACLE's z_i > s_i decision is meaningful: not a near-tie, and not a degenerate pair where
    BOTH z_i and s_i have converged to zero (a contact exactly at the boundary: which of two 1e-9 numbers is larger
    is decided by the rounding of the last PDIPM iteration).  Index sets are compared bit-exactly on these rows; the
    share of rows this mask drops is reported and gated (MAX_MASKED_FRAC) so that the mask cannot grow silently."""
    big = torch.maximum(z.abs(), s.abs())
    zs = z.abs().max(dim=1, keepdim=True)[0]
    ss = s.abs().max(dim=1, keepdim=True)[0]
    nondegenerate = torch.maximum(z.abs() / zs, s.abs() / ss) > floor
    return ((z - s).abs() > rel * big) & nondegenerate


REORDER_PROBES = 16            # oracle solves with its unknowns in another order, per report, on the scenes with a row nothing else explains
MAX_MASKED_FRAC = 0.02          # index-set rows the decisive_rows mask may drop on the BASELINE stack configs
MIN_WELL_POSED_FRAC = 0.9       # scenes whose backward system the oracle itself solves (stack configs)


def backward_well_posed(Q, G, A, F, ref, cot, gref):
    """Scenes where the oracle's own backward (lcp.py:44-50) is a solved system: its KKT residual is small (scenes that
    over-converged, mu ~ 1e-17, leave a matrix singular to fp64 working precision) and the solution is strictly
    complementary (a pair with z_i ~ s_i ~ 0 makes the gradient one-sided).  `gref` holds dp, dh, db."""
    res = kkt_backward_residual(Q, G, A, F, ref.z, ref.s, cot, gref["dp"], -gref["dh"],
                                None if gref.get("db") is None else -gref["db"])
    ok = torch.stack([v for v in res.values()]).max(dim=0)[0] < 1e-9
    zs, ss = ref.z.max(dim=1, keepdim=True)[0], ref.s.max(dim=1, keepdim=True)[0]
    ok = ok & (torch.maximum(ref.z / zs, ref.s / ss).min(dim=1)[0] > 1e-6)
    # ... and its dlam has the size the data allow: on a solve that converged to rounding (mu ~ 1e-15) the oracle's own solve can return
    # dlam ~ 1e16 along a null direction of its singular matrix with a RELATIVE residual of 1e-17 (measured: configs[1] with a coupled
    # equality row, scene 487 of seed 1236: |dlam| = 7.6e16, natural scale ~10; tools/experiments/bwd_outliers.py) - an answer no other
    # elimination order reproduces and no gradient anybody could use
    rown = G.double().norm(dim=2)                                                   # |G_i|; zero rows (the gamma rows) do not constrain dlam
    gmin = torch.where(rown > 0, rown, torch.full_like(rown, float("inf"))).min(dim=1)[0].clamp(1e-300, 1e300)
    sane = _n(gref["dh"]) <= 1e8 * _n(cot) / gmin                                   # (Q dx + G^T dlam = -cot: |dlam| ~ |cot| / |G_i| by nature)
    return ok & sane


# ----------------------------------------------------------------------------
# backward parity that is well-posed on degenerate contact LCPs
# ----------------------------------------------------------------------------
# With two opposite friction directions per contact (world.py:191-192) a sticking
# contact has BOTH friction multipliers active, and only their difference is determined
# by the limiting KKT system; the (1,1) component of dlam is fixed by diag(s/z) ~ 1e-17
# entries, i.e. by rounding.  The reference's own fp64 result is therefore not
# reproducible in dG/dh/dF by any other elimination order (measured: O(1e-1) between
# the reference and an algebraically identical block solve), while dp, dQ, dA, db and
# every gradient w.r.t. a *physical* parameter (where the +/- friction rows are tied
# together by the assembly) agree to rounding.  Hence three checks:
#   1. direct scaled parity on dp, dQ, dA, db (always), all 7 on non-degenerate LCPs;
#   2. the KKT residual of (dx, dlam, dnu) in the system the reference solves;
#   3. parity of the gradients contracted through the assembly (`physical_grads`).


def kkt_backward_residual(Q, G, A, F, z, s, cot, dx, dlam, dnu=None):
    """Row-block scaled residuals of lcp.py:47-50's system, ds eliminated (ds = -dz/d):
         Q dx + G^T dlam + A^T dnu = -cot ;  G dx - dlam/d - F dlam = 0 ;  A dx = 0.
    ||dx|| is floored by the free-motion ||Q^-1 cot|| (a fully constrained dx is noise)."""
    D = lambda t: None if t is None else t.double()
    Q, G, A, F, z, s, cot, dx, dlam, dnu = map(D, (Q, G, A, F, z, s, cot, dx, dlam, dnu))
    mv = lambda M, v: torch.bmm(M, v.unsqueeze(-1)).squeeze(-1)
    mtv = lambda M, v: torch.bmm(M.transpose(1, 2), v.unsqueeze(-1)).squeeze(-1)
    ndx = torch.maximum(_n(dx), _n(torch.linalg.solve(Q, cot.unsqueeze(-1))))
    r1 = mv(Q, dx) + mtv(G, dlam) + cot
    den1 = _n(Q) * ndx + _n(G) * _n(dlam) + _n(cot)
    if A is not None:
        r1 = r1 + mtv(A, dnu)
        den1 = den1 + _n(A) * _n(dnu)
    sd = (s / z) * dlam
    r3 = mv(G, dx) - sd - mv(F, dlam)
    den3 = _n(G) * ndx + _n(sd) + _n(F) * _n(dlam)
    out = {"dual": _n(r1) / den1.clamp_min(1e-300), "ineq": _n(r3) / den3.clamp_min(1e-300)}
    if A is not None:
        out["eq"] = _n(mv(A, dx)) / (_n(A) * ndx).clamp_min(1e-300)
    return out


PHYS_KEYS = ["Mdiag", "v", "f", "c_n", "c_p1", "c_p2", "rest", "fric"]


def physical_grads(phys, dt, grads, oracle):
    """Contract LCP-level grads (dict Q,p,G,h,F) through the engine assembly
    (engines.py:31-32,50-74; world.py:144-234) with autograd of the oracle's restatement.
    phys: dict with PHYS_KEYS (+ c_i1, c_i2, Je), batched."""
    ins = {k: phys[k].double().clone().requires_grad_(True) for k in PHYS_KEYS}
    Je = phys.get("Je")
    Q, p, G, h, A, b, F = oracle.assemble_lcp(
        ins["Mdiag"], ins["v"], ins["f"], dt, ins["c_n"], ins["c_p1"], ins["c_p2"],
        phys["c_i1"], phys["c_i2"], ins["rest"], ins["fric"],
        None if Je is None else Je.double())
    outs = [Q, p, G, h, F]
    gos = [grads[k].double() for k in "QpGhF"]
    g = torch.autograd.grad(outs, [ins[k] for k in PHYS_KEYS], gos, allow_unused=True)
    return {k: (torch.zeros_like(ins[k]) if gi is None else gi) for k, gi in zip(PHYS_KEYS, g)}


def err_physical(pg, pg_ref, phys, floor, keys=None):
    """Joint error of d(loss)/d(log-ish theta): blocks weighted by max(||theta||, 1).

    `keys` restricts the comparison.  With redundant contact points (more than two collinear points
    on one box-box interface, as in the 4-points-per-interface BASELINE shapes) the normal and the
    friction multipliers - and with them dlam - are non-unique along the null space of the active
    Jacobian rows, so gradients w.r.t. contact geometry / friction are not defined even for the
    reference; only parameters that enter through Q and p (masses, velocities, forces) are."""
    num = 0.0
    den = 0.0
    for k in (keys or PHYS_KEYS):
        w = _n(phys[k]).clamp_min(1.0)
        num = num + (w * _n(pg[k] - pg_ref[k])) ** 2
        den = den + (w * _n(pg_ref[k])) ** 2
    return num.sqrt() / torch.maximum(den.sqrt(), floor).clamp_min(1e-300)


def own_iterate_backward(oracle, lcp64, ref, cot, x, z, s, grads, fl, wobble=1e-7, moved_tol=1e-5, seed=11):
    """The kernel's backward against the ORACLE'S BACKWARD EVALUATED AT THE KERNEL'S OWN ITERATE - the same linear system (lcp.py:44-50), two
    solvers - on every scene where that system determines its solution.  Round 5 (a gate that does not depend on which
    iterate either side kept, and does not select on the outcome): the conditioning test uses the oracle alone - the system at (z, s) is
    well posed by the oracle's own residual (`backward_well_posed`) and its dl/dp, dQ, db move by less than `moved_tol` when (z, s) are
    moved by a relative `wobble`.  A residual alone cannot see a solve that divided by rounding noise (multipliers of 1e15 satisfy the
    system to 1e-16 of their own size: tools/experiments/fp64_io_backward_diag.py).  Compared: dl/dp, dQ, db (dA needs the kernel's nu,
    which the metric's entry points do not return).  Fields: bwd_own_iterate_determined_scenes, bwd_own_iterate_err_max / _median."""
    import copy
    Q, p, G, h, A, b, F = lcp64

    def at_iterate(zz, ss):
        at = copy.copy(ref)
        at.x, at.z, at.s = x, zz, ss                      # (y: the oracle's - only dA reads it)
        g = oracle.lcp_backward(at, *lcp64, cot)
        return g, backward_well_posed(Q, G, A, F, at, cot, g)

    keys = [k for k in ("p", "Q", "b") if grads.get(k) is not None]
    g1, ok1 = at_iterate(z, s)
    gen = torch.Generator().manual_seed(seed)
    wob = lambda t: t * (1 + wobble * torch.randn(t.shape, generator=gen, dtype=torch.float64))
    g2, ok2 = at_iterate(wob(z), wob(s))
    moved = err_grads({k: g2["d" + k] for k in keys}, {k: g1["d" + k] for k in keys}, fl)
    det = ok1 & ok2 & (torch.stack(list(moved.values())).max(dim=0)[0] < moved_tol)
    errs = err_grads({k: grads[k].double() for k in keys}, {k: g1["d" + k] for k in keys}, fl)
    worst = torch.stack(list(errs.values())).max(dim=0)[0]
    out = {"bwd_own_iterate_determined_scenes": int(det.sum()), "bwd_own_iterate_compared": "d" + ", d".join(keys)}
    if bool(det.any()):
        out["bwd_own_iterate_err_max"] = float(worst[det].max())
        out["bwd_own_iterate_err_median"] = float(worst[det].median())
    return out


# ----------------------------------------------------------------------------
# one report for "the kernel the metric times against the oracle" (tests/test_hip_headline_parity.py and
# bench.py's `parity` object print the same fields)
# ----------------------------------------------------------------------------
def headline_report(oracle, lcp64, x, z, s, iters, dp=None, cot=None, floors=(1e-5, 1e-4), grads=None, phys=None, dt=None,
                    phys_grads=None, input_stability=False, all_grads=False, cache=None, oracle_stability="if_needed"):
    """`lcp64`: the LCP the kernel solved (fp64 copies of the fp32 data the HIP assembly produced: identical inputs),
    `x, z, s, iters`: what the kernel returned for those scenes, `dp` (optional): its dl/dp for the cotangent `cot`.
    `grads` (optional): the kernel's dense gradients, dict over "QpGhAbF" (lcp.py:52-61); `phys` + `dt`: the scenes' physical
    description (`SceneBatch.phys_dict()`, CPU) - the dense gradients are then also contracted through the engine assembly and
    compared there; `phys_grads`: gradients w.r.t. the physical inputs the kernel returned DIRECTLY (`lcp_step_backward_f32`,
    dict over PHYS_KEYS) - compared with the oracle's dense gradients contracted through the assembly, and dl/dp is read off
    d(loss)/df = dt dl/dp (engines.py:32: p = M v + dt f).
    Returns (report dict, oracle solution).  Fields:
      fwd_err_x_*                      SURVEY 8d err_x per scene
      index_set_mismatches_unmasked    rows where (z_i > s_i) differs from the oracle's, NO mask
      index_set_mismatches             the same on the decisive rows (`decisive_rows`, floor = floors[0])
      index_set_masked_frac            share of rows that mask drops (ties of two numbers that both converged to zero)
      index_set_mismatches_floor_1e-4 / masked_frac_floor_1e-4   with the looser floor the body-space kernels are gated on
      index_set_mismatches_oracle_stable   differing rows whose ORACLE decision is stable under fp32 arithmetic AND the pivot-free LU
                                       (see the code: gated to zero; `index_set_mismatch_rows` lists the first 64 differing rows)
      ref_fp32_vs_fp64_err_x_max / _index_rows   SURVEY 8(d)'s companion: the reference algorithm's own fp32-vs-fp64 error on these inputs
                                       (`oracle_stability`: "always" computes them, "if_needed" only when a row differs, "never")
      iters_delta_hist                 histogram of iters - iters_oracle (pdipm.py:80-136 loop iterations per scene)
      iters_differ_frac                share of scenes with a non-zero delta
      bwd_err_dp_*                     dl/dp against lcp.py:52 on the scenes whose backward system is well posed
      bwd_err_dQ_max / dA / db         the other outputs of lcp.py:52-61 that are defined on degenerate contact LCPs (see the note
                                       above `kkt_backward_residual`), same scenes, `grad_floors` scaling
      bwd_kkt_resid_max                residual of the kernel's (dx, dlam, dnu) = (dp, -dh, -db) in the system lcp.py:47-50 solves at
                                       the iterate the KERNEL returned (its fp32 outputs z, s; the iterate itself is what the forward
                                       fields compare); bwd_kkt_resid_at_oracle_iterate_max: the same vectors in the system at the
                                       ORACLE's iterate - informative only: where a solve converged to rounding s_i / z_i is a ratio of two
                                       1e-17 numbers and the two systems are different matrices (configs[1]: O(1))
      bwd_err_phys_max                 `err_physical` over Mdiag, v, f (the parameters that enter through Q and p)
    `input_stability` (needs `phys`, `dt`): the backward fields are taken over the well-posed scenes whose ORACLE answer is itself stable
    under fp32 rounding of its inputs - the oracle is run a second time on its own fp64 assembly of the physical inputs (the dense
    tensors in `lcp64` are fp32 numbers) and a scene whose oracle dl/dp moves by more than the tolerance between the two is counted in
    `bwd_input_sensitive_scenes` instead (`bwd_input_sensitivity_max`: the largest such move).  One scene of the 32768 of BASELINE
    configs[3] is (shard 4, scene 146: its oracle moves by 5e-6 in x and 1e-3 in dl/dp - profiles/r04_config3_all_shards_parity.json).
    On EVERY scene, well-posed or not (round 5): `bwd_nonfinite_scenes` - scenes with a NaN / inf anywhere in the gradients the kernel
    returned - and, with dense gradients, `bwd_kkt_resid_all_max`: the residual of (dx, dlam, dnu) = (dp, -dh, -db) in lcp.py:47-50's system
    at the iterate the kernel returned, over all scenes (`bwd_kkt_resid_all_scenes_over_1e-6`: how many exceed 1e-6).  That system is the
    kernel's own - no oracle solution enters -, so it holds the scenes the filters above drop to "the kernel solved what lcp.py:44-50 poses".
    `all_grads`: dG, dh, dF are compared too (`bwd_err_dG_max` ...: informative, see the note at `bwd_err_dlam_determined_max` in the code)
    and the physical gradients over ALL of PHYS_KEYS (`bwd_err_phys_all_max`): meaningful where the multipliers are unique - ONE point per
    interface; the reference's own two points per interface are already redundant in the tangential direction (both lie on the line their
    friction directions span: measured, the split of the friction gradient between the two points of an interface differs O(1)
    between any two solves while its sum agrees).
    With BOTH `grads` and `phys_grads` the physical comparison is reported for each source (`bwd_err_phys_max` = the dense gradients
    contracted through the assembly, `bwd_err_phys_direct_max` = lcp_step_backward_f32's own outputs).
    `cache` (a dict the caller keeps): the oracle's forward / backward of these LCPs is stored there and re-used by the next call with
    the same dict (several entry points solving the same scenes)."""
    Q, p, G, h, A, b, F = lcp64
    cache = {} if cache is None else cache
    if "ref" not in cache:
        cache["ref"] = oracle.lcp_forward(*lcp64)
    ref = cache["ref"]
    n = Q.shape[0]
    ex = err_x(x.double(), ref.x, Q, p)
    out = {"scenes": int(n), "tolerance": 1e-4, "fwd_err_x_max": float(ex.max()), "fwd_err_x_median": float(ex.median())}
    z, s = z.double(), s.double()
    diff = (z > s) != (ref.z > ref.s)
    out["index_set_rows_total"] = int(diff.numel())
    out["index_set_mismatches_unmasked"] = int(diff.sum())
    for i, fl in enumerate(floors):
        dec = decisive_rows(ref.z, ref.s, floor=fl)
        sfx = "" if i == 0 else "_floor_%g" % fl
        out["index_set_mismatches" + sfx] = int((diff & dec).sum())
        out["index_set_masked_frac" + sfx] = 1.0 - float(dec.sum()) / dec.numel()
        if i == 0:
            out["index_set_rows_compared"] = int(dec.sum())
    # Every row that differs from the oracle's, accounted for (north_star: "contact index sets bit-exact"): is the
    # oracle's own decision z_i > s_i on that row stable?  It is solved again (a) in fp32 arithmetic on the same numbers (the reference's
    # dtype on a float32 world: pdipm.py runs in the dtype of its inputs) and (b) with the pivot-free LU the reference takes on its GPU
    # path (pdipm.py:15-28 `btrifact_hack`): a row either of them flips is a row the reference itself does not decide reproducibly.
    # `index_set_mismatches_oracle_stable` counts the differing rows that none of (a) .. (d) below explains - gated to zero everywhere.
    # The same two solves give SURVEY 8(d)'s noise-floor companion: the reference algorithm's own fp32-vs-fp64 error on these inputs.
    if oracle_stability == "always" or (oracle_stability == "if_needed" and int(diff.sum()) > 0):
        if "ref_f32" not in cache:
            cache["ref_f32"] = oracle.lcp_forward(*[None if t is None else t.float() for t in lcp64])
            cache["ref_nopivot"] = oracle.lcp_forward(*lcp64, pivot=False)
        r32, rnp = cache["ref_f32"], cache["ref_nopivot"]
        flip_a = (r32.z.double() > r32.s.double()) != (ref.z > ref.s)
        flip_b = (rnp.z > rnp.s) != (ref.z > ref.s)
        # rows neither of the two flips: two more looks at the ORACLE alone, on the scenes concerned -
        # (c) the same LCP with its unknowns and rows in another order (up to REORDER_PROBES permutation similarities: identical in exact
        #     arithmetic, rounded differently - what another host's BLAS does to the oracle; measured: the GPU box's host and this container's keep different
        #     iterates on scenes 117 / 330 of configs[1]);
        # (d) its own trajectory (trace): the iterate it KEPT is chosen by `resid < best` and by the exact-zero-pivot exit of
        #     pdipm.py:99-102 - on a converged solve both are decided by the last bits - and one iteration on the complementary products
        #     shrink ~1000 x: a pair (z_i, s_i) that both go to zero changes sides.  A differing row is explained when the kernel decides it as the
        #     oracle itself does at a NEIGHBOUR (at most two passes away) of the iterate it kept, whose x equals the kept one's to 1e-6 of the
        #     free motion; "neighbour" includes the pass the oracle would have taken had it not stopped on 0.25 eps <= resid < eps.
        flip_c = torch.zeros_like(diff)
        via_iterate = torch.zeros_like(diff)
        iterate_note = {}
        scn = torch.nonzero((diff & ~(flip_a | flip_b)).any(dim=1)).flatten()
        tr, tr_more = [], []
        if scn.numel():
            sub = [None if t is None else t[scn] for t in lcp64]
            nzs, ms = sub[0].shape[1], sub[2].shape[1]
            nes = 0 if sub[4] is None else sub[4].shape[1]
            base_dec = ref.z[scn] > ref.s[scn]
            kdec_s = z[scn] > s[scn]
            need = (diff & ~(flip_a | flip_b))[scn]
            xfree = torch.maximum(free_scales(sub[0], sub[1])["x_free"], ref.x[scn].norm(dim=1)).clamp_min(1e-300)
            # the oracle on these scenes alone, with its trajectory, and continued past `best_resid < eps` (pdipm.py:133; eps = 1e-12: a residual of
            # ~1e-12 is a sum of differences of O(1) numbers, known to a few per cent - measured: 8.96e-13 alone, 9.17e-13 in a batch of three):
            # the extra passes count only for scenes that stopped with 0.25 eps <= resid < eps
            rs_ = oracle.lcp_forward(*sub, trace=tr)
            oracle.lcp_forward(*sub, trace=tr_more, eps=0.0)
            window = (rs_.resid >= 0.25e-12) & (rs_.resid < 1e-12)
            fc = torch.zeros_like(need)                  # rows the reordered oracle flips at the iterate it keeps
            seen = torch.zeros_like(need)                # rows on which SOME iterate (same solution to 1e-6) of SOME probe decides as the kernel does
            where = {}

            def look(trace, zsel, xsel, tag, allowed):
                for j, t in enumerate(trace):
                    zz, ss, xx = zsel(t["z"].double()), zsel(t["s"].double()), xsel(t["x"].double())
                    close = ((xx - ref.x[scn]).norm(dim=1) / xfree <= 1e-6) & allowed
                    hit = ((zz > ss) == kdec_s) & close.unsqueeze(1) & need & ~seen
                    for i in torch.nonzero(hit.any(dim=1)).flatten().tolist():
                        where.setdefault(int(scn[i]), {"probe": tag, "oracle_iteration": j, "oracle_resid_there": float(t["resid"][i]),
                                                       "oracle_iterations_run": int(rs_.iters[i]), "oracle_resid_kept": float(rs_.resid[i])})
                    seen.__ior__(hit)

            ident = lambda v: v
            every = torch.ones(scn.numel(), dtype=torch.bool)
            look(tr, ident, ident, "its own trajectory", every)
            look(tr_more, ident, ident, "its own trajectory continued past resid < eps", window)
            probes = 0
            for seed in range(REORDER_PROBES):           # the first probe reverses every order, the others are random permutations
                if not bool((need & ~(fc | seen)).any()):
                    break
                g_ = torch.Generator().manual_seed(seed)
                px = torch.arange(nzs - 1, -1, -1) if seed == 0 else torch.randperm(nzs, generator=g_)
                pm = torch.arange(ms - 1, -1, -1) if seed == 0 else torch.randperm(ms, generator=g_)
                pe = None if nes == 0 else (torch.arange(nes - 1, -1, -1) if seed == 0 else torch.randperm(nes, generator=g_))
                Ap, bp = (None, None) if nes == 0 else (sub[4][:, pe][:, :, px], sub[5][:, pe])
                args = (sub[0][:, px][:, :, px], sub[1][:, px], sub[2][:, pm][:, :, px], sub[3][:, pm], Ap, bp, sub[6][:, pm][:, :, pm])
                ipm, ipx = torch.argsort(pm), torch.argsort(px)
                trp, trp_more = [], []
                rp = oracle.lcp_forward(*args, trace=trp)
                fc |= (rp.z[:, ipm] > rp.s[:, ipm]) != base_dec
                look(trp, lambda v: v[:, ipm], lambda v: v[:, ipx], "reordered (probe %d)" % seed, every)
                if bool(window.any()):
                    oracle.lcp_forward(*args, trace=trp_more, eps=0.0)
                    look(trp_more, lambda v: v[:, ipm], lambda v: v[:, ipx], "reordered (probe %d), continued past resid < eps" % seed, window)
                probes += 1
            flip_c[scn] = fc
            via_iterate[scn] = seen
            iterate_note = where
            out["index_set_reorder_probes"] = probes
        explained = flip_a | flip_b | flip_c | via_iterate
        if scn.numel():
            left = [int(v) for v in torch.nonzero((diff & ~explained).any(dim=1)).flatten()[:8]]
            pos = {int(v): k for k, v in enumerate(scn.tolist())}
            out["index_set_unexplained_scenes"] = {str(v): {"oracle_iters": int(ref.iters[v]), "kernel_iters": int(iters[v]), "oracle_resid_kept": float(ref.resid[v]),
                                                           "oracle_resid_by_iteration": [float(t["resid"][pos[v]]) for t in tr],
                                                           "oracle_resid_continued_past_eps": [float(t["resid"][pos[v]]) for t in tr_more]} for v in left}
        out["index_set_mismatches_oracle_stable"] = int((diff & ~explained).sum())
        out["index_set_mismatches_oracle_flips_in_fp32"] = int((diff & flip_a).sum())
        out["index_set_mismatches_oracle_flips_without_pivoting"] = int((diff & flip_b).sum())
        out["index_set_mismatches_oracle_flips_reordered"] = int((diff & flip_c).sum())
        out["index_set_mismatches_at_another_oracle_iterate"] = int((diff & via_iterate & ~(flip_a | flip_b | flip_c)).sum())
        out["index_set_mismatch_scenes_at_another_oracle_iterate"] = {str(k): v for k, v in list(iterate_note.items())[:16]}
        rows = torch.cat([torch.nonzero(diff & ~explained), torch.nonzero(diff & explained)])       # (the unexplained ones first)
        out["index_set_mismatch_rows"] = [
            {"scene": int(i), "row": int(r), "z": float(z[i, r]), "s": float(s[i, r]), "z_oracle": float(ref.z[i, r]), "s_oracle": float(ref.s[i, r]),
             "oracle_flips_in_fp32": bool(flip_a[i, r]), "oracle_flips_without_pivoting": bool(flip_b[i, r]), "oracle_flips_reordered": bool(flip_c[i, r]),
             "kernel_set_is_another_oracle_iterate": bool(via_iterate[i, r])} for i, r in rows[:64].tolist()]
        e32 = err_x(r32.x.double(), ref.x, Q, p)
        out["ref_fp32_vs_fp64_err_x_max"] = float(e32.max())
        out["ref_fp32_vs_fp64_err_x_median"] = float(e32.median())
        out["ref_fp32_vs_fp64_index_rows"] = int(flip_a.sum())
        out["ref_nopivot_vs_pivot_err_x_max"] = float(err_x(rnp.x, ref.x, Q, p).max())
        out["ref_nopivot_vs_pivot_index_rows"] = int(flip_b.sum())
    else:
        out["index_set_mismatches_oracle_stable"] = 0          # (no differing row at all)
    d = (iters.to(torch.int64).cpu() - ref.iters.to(torch.int64))
    out["iters_delta_hist"] = {str(int(k)): int((d == k).sum()) for k in torch.unique(d)}
    out["iters_differ_frac"] = float((d != 0).sum()) / n
    out["iters_max_abs_delta"] = int(d.abs().max())
    if grads is not None and dp is None:
        dp = grads["p"]
    if phys_grads is not None and dp is None:
        dp = phys_grads["f"].double().reshape(n, -1) / dt              # d(loss)/df = dt dl/dp
    if dp is not None:
        c64 = cot.double()
        if "gref" not in cache:
            cache["gref"] = oracle.lcp_backward(ref, *lcp64, c64)
        gref = cache["gref"]
        # every scene, before any filter: finite gradients, and the kernel's own backward system solved
        fin = torch.ones(n, dtype=torch.bool)
        for gsrc in (grads, phys_grads):
            for t in (gsrc or {}).values():
                if t is not None:
                    fin &= torch.isfinite(t.reshape(n, -1).double()).all(dim=1)
        out["bwd_nonfinite_scenes"] = int((~fin).sum())
        if grads is not None and grads.get("h") is not None:
            dnu_k = None if (A is None or grads.get("b") is None) else -grads["b"].double()
            res_all = kkt_backward_residual(Q, G, A, F, z, s, c64, grads["p"].double(), -grads["h"].double(), dnu_k)
            worst = torch.stack([torch.nan_to_num(v, nan=float("inf")) for v in res_all.values()]).max(dim=0)[0]
            out["bwd_kkt_resid_all_max"] = float(worst.max())
            out["bwd_kkt_resid_all_scenes_over_1e-6"] = int((worst > 1e-6).sum())
            out["bwd_kkt_resid_all_median"] = float(worst.median())
        ok = backward_well_posed(Q, G, A, F, ref, c64, gref)
        fl = grad_floors(Q, p, c64, ref.x, ref.z, ref.y)
        # (with the physical gradients alone - lcp_step_backward_f32 - dl/dp = d(loss)/df / dt is what there is to compare)
        out.update(own_iterate_backward(oracle, lcp64, ref, c64, x.double(), z, s, grads if grads is not None else {"p": dp.double()}, fl))
        if input_stability:
            ph64 = {k: (v.double() if (v is not None and v.is_floating_point()) else v) for k, v in phys.items()}
            if "ref_o" not in cache:
                lcp_o = oracle.assemble_lcp(ph64["Mdiag"], ph64["v"], ph64["f"], dt, ph64["c_n"], ph64["c_p1"], ph64["c_p2"], ph64["c_i1"], ph64["c_i2"],
                                            ph64["rest"], ph64["fric"], ph64.get("Je"))
                cache["ref_o"] = oracle.lcp_forward(*lcp_o)
                cache["g_o"] = oracle.lcp_backward(cache["ref_o"], *lcp_o, c64)
            ref_o, g_o = cache["ref_o"], cache["g_o"]
            sens = err_grads({"p": g_o["dp"]}, {"p": gref["dp"]}, fl)["p"]
            stable = sens <= out["tolerance"]
            out["bwd_input_sensitive_scenes"] = int((ok & ~stable).sum())
            out["bwd_input_sensitivity_max"] = float(sens[ok].max()) if bool(ok.any()) else 0.0
            out["fwd_input_sensitivity_err_x_max"] = float(err_x(ref_o.x, ref.x, Q, p).max())
            ok = ok & stable
        eg = err_grads({"p": dp.double()}, {"p": gref["dp"]}, fl)["p"]
        if bool(ok.any()):           # (callers gate bwd_well_posed_frac; without a well-posed scene the bwd_err_* keys are ABSENT and a gate on them fails loudly)
            out.update({"bwd_err_dp_max": float(eg[ok].max()), "bwd_err_dp_median": float(eg[ok].median())})
        out["bwd_well_posed_scenes"] = int(ok.sum())
        out["bwd_well_posed_frac"] = float(ok.sum()) / n
        gr = {k: gref["d" + k] for k in "QpGhAbF"}
        if grads is not None and bool(ok.any()):
            g64 = {k: (None if grads.get(k) is None else grads[k].double()) for k in "QpGhAbF"}
            keys = [k for k in ("QAbGhF" if all_grads else "QAb") if g64.get(k) is not None and gr[k] is not None]
            errs = err_grads({k: g64[k] for k in keys}, {k: gr[k] for k in keys}, fl)
            for k in keys:
                out["bwd_err_d%s_max" % k] = float(errs[k][ok].max())
            if all_grads:
                # dG, dh, dF are functions of dlam = -dh, and of dlam only a part is DETERMINED on a contact LCP: a sticking contact has both
                # friction multipliers active, (dlam_f1 - dlam_f2) enters G^T dlam and the limiting system fixes nothing else about the pair
                # (measured, one point per interface: the sum differs by up to 1e3 between kernel and oracle where the difference agrees to
                # 1e-7).  Compared: (dlam_n, dlam_f1 - dlam_f2, dlam_gamma) on every well-posed scene, the rows of dG they determine, and the
                # full dh / dG / dF on the scenes WITHOUT a sticking contact (z_gamma > s_gamma on every contact), which are counted.
                nc_ = G.shape[1] // 4
                det = lambda dl: torch.cat([dl[:, :nc_], dl[:, nc_:3 * nc_:2] - dl[:, nc_ + 1:3 * nc_:2], dl[:, 3 * nc_:]], 1)
                dlk, dlo = -g64["h"], -gr["h"]
                e_det = _n(det(dlk) - det(dlo)) / torch.maximum(_n(det(dlo)), fl["h"]).clamp_min(1e-300)
                out["bwd_err_dlam_determined_max"] = float(e_det[ok].max())
                detG = lambda dG_: torch.cat([dG_[:, :nc_], dG_[:, nc_:3 * nc_:2] - dG_[:, nc_ + 1:3 * nc_:2], dG_[:, 3 * nc_:]], 1)
                e_dG = _n(detG(g64["G"]) - detG(gr["G"])) / torch.maximum(_n(detG(gr["G"])), fl["G"]).clamp_min(1e-300)
                out["bwd_err_dG_determined_max"] = float(e_dG[ok].max())
                slide = ok & ~(ref.z[:, 3 * nc_:] < ref.s[:, 3 * nc_:]).any(dim=1)
                out["bwd_scenes_without_sticking_contact"] = int(slide.sum())
                if bool(slide.any()):
                    for k in "hGF":
                        out["bwd_err_d%s_no_sticking_max" % k] = float(errs[k][slide].max())
            dnu = None if (A is None or g64.get("b") is None) else -g64["b"]
            for name, zz, ss in (("bwd_kkt_resid_max", z, s), ("bwd_kkt_resid_at_oracle_iterate_max", ref.z, ref.s)):
                res = kkt_backward_residual(Q, G, A, F, zz, ss, c64, g64["p"], -g64["h"], dnu)
                out[name] = max(float(v[ok].max()) for v in res.values())
        if phys is not None and bool(ok.any()) and (grads is not None or phys_grads is not None):
            ph = {k: (v.double() if (v is not None and v.is_floating_point()) else v) for k, v in phys.items()}
            keys = ["Mdiag", "v", "f"]
            pg_ref = physical_grads(ph, dt, gr, oracle)
            scl = free_scales(Q, p, c64)
            floor = _n(c64) * torch.maximum(scl["x_free"], _n(ref.x))
            sources = []
            if grads is not None:
                sources.append(("", physical_grads(ph, dt, {k: g64[k] for k in "QpGhF"}, oracle),
                                "the kernel's dense gradients contracted through the assembly (engines.py:31-32,50-74)"))
            if phys_grads is not None:
                sources.append(("_direct" if grads is not None else "", {k: phys_grads[k].double() for k in PHYS_KEYS if k in phys_grads},
                                "the kernel's own physical gradients (lcp_step_backward_f32)"))
            for sfx, pg, what in sources:
                ep = err_physical(pg, pg_ref, ph, floor, keys=keys)
                out["bwd_err_phys%s_max" % sfx] = float(ep[ok].max())
                out["bwd_err_phys%s_median" % sfx] = float(ep[ok].median())
                out["bwd_phys%s_source" % sfx] = what
                if all_grads:
                    epa = err_physical(pg, pg_ref, ph, floor, keys=PHYS_KEYS)
                    out["bwd_err_phys_all%s_max" % sfx] = float(epa[ok].max())
                    out["bwd_err_phys_all%s_median" % sfx] = float(epa[ok].median())
                    # the five parameters whose gradients are functions of (dx, dlam_n, dlam_f1 - dlam_f2, dlam_gamma) alone - determined
                    # also where the SPLIT of a friction pair or between two points of an interface is not (restitution enters h's normal
                    # rows, friction F's (gamma, n) entries: engines.py:61-74): Mdiag, v, f, rest, fric
                    ep5 = err_physical(pg, pg_ref, ph, floor, keys=["Mdiag", "v", "f", "rest", "fric"])
                    out["bwd_err_phys_five%s_max" % sfx] = float(ep5[ok].max())
    return out, ref

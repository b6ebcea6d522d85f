"""CPU: the reciprocal form of get_step the fp64 kernels evaluate (lcp_device.h: step_flags / step_from_flags; LCP_Q_RCP_STEP,
LCP_SOLO_RCP_STEP, LCP_PRIMAL_RCP_STEP) is the reference's get_step (pdipm.py:182-186), case for case.

The kernels form t_i = dv_i (1 / v_i) instead of the quotients a_i = -v_i / dv_i = -1 / t_i.  This file restates what they do with
the t_i in numpy and holds it against the oracle's `get_step` on adversarial vectors: exact zeros of either sign, NaN and infinities
among the dv_i, vectors without a decreasing entry (the fill max(1, a.max()) IS the result then), vectors without an increasing one."""
import numpy as np
import torch

from oracle import pdipm_oracle as O


def step_rcp_model(v, dv):
    with np.errstate(all="ignore"):
        t = dv * (1.0 / v)
        nan = np.isnan(t).any(1)                           # step_flags bit 0
        pz = ((t == 0) & ~np.signbit(t)).any(1)            # bit 1: an exact +0 (a_i = -inf)
        nz = ((t == 0) & np.signbit(t)).any(1)             # bit 2: an exact -0 (a_i = +inf)
        neg = (t < 0).any(1)                               # bit 3: a decreasing entry
        pos = (t > 0).any(1)                               # bit 4: an entry the fill replaces
        tmin = np.fmin.reduce(t, axis=1)                   # NaN-ignoring minimum (v_min_f64)
        # step_from_flags
        return np.where(nan, np.nan, np.where(pz, -np.inf, np.where(neg, -1.0 / tmin, np.where(pos, np.where(nz, np.inf, 1.0), np.inf))))


def _vectors(seed, N=60000, m=8, special_frac=0.15):
    rng = np.random.default_rng(seed)
    v = np.exp(rng.normal(0, 3, (N, m)))
    dv = rng.normal(0, 1, (N, m)) * np.exp(rng.normal(0, 3, (N, m)))
    special = np.array([np.nan, 0.0, -0.0, np.inf, -np.inf])
    mask = rng.random((N, m)) < special_frac
    dv[mask] = special[rng.integers(0, 5, int(mask.sum()))]
    k = N // 6
    dv[:k] = np.abs(dv[:k])                                # no decreasing entry: the fill is the result
    dv[k:2 * k] = -np.abs(dv[k:2 * k])                     # no increasing entry
    return v, dv


def test_reciprocal_form_equals_get_step_case_for_case():
    for seed, frac in ((0, 0.15), (1, 0.0), (2, 0.5)):
        v, dv = _vectors(seed, special_frac=frac)
        ref = O.get_step(torch.from_numpy(v), torch.from_numpy(dv)).numpy()
        got = step_rcp_model(v, dv)
        assert (np.isnan(ref) == np.isnan(got)).all()
        fin = ~np.isnan(ref)
        eq = got[fin] == ref[fin]                          # (covers the infinities)
        with np.errstate(all="ignore"):
            rel = np.abs(got[fin] - ref[fin]) / np.abs(ref[fin])
        assert (eq | (rel < 1e-15)).all(), float(np.nanmax(rel[~eq]))
        # every branch of step_from_flags was taken
        assert np.isnan(ref).any() and (ref == -np.inf).any() and (ref == np.inf).any() and (ref == 1.0).any() or frac == 0.0


def test_pair_of_vectors_combines_like_the_reference():
    """alpha = min(get_step(z, dz), get_step(s, ds)) (pdipm.py:142-144, :164-166): the NaN-propagating minimum of the two vectors' steps;
    where both have a decreasing entry and nothing is special it is -1 / min over BOTH vectors' t_i (the kernels' fast form)."""
    v1, d1 = _vectors(5, N=20000, special_frac=0.0)
    v2, d2 = _vectors(6, N=20000, special_frac=0.0)
    a = O.get_step(torch.from_numpy(v1), torch.from_numpy(d1))
    b = O.get_step(torch.from_numpy(v2), torch.from_numpy(d2))
    ref = torch.minimum(a, b).numpy()
    t = np.concatenate([d1 / v1, d2 / v2], axis=1)
    both = ((d1 < 0).any(1)) & ((d2 < 0).any(1))
    fast = -1.0 / t.min(axis=1)
    assert np.allclose(fast[both], ref[both], rtol=1e-14, atol=0)

"""Compatibility shim that lets the UNMODIFIED reference run in this container.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file.  It
is used (a) by `oracle/make_golden.py` to generate the committed fixtures under
`tests/golden/` and (b) by CPU tests that are skipped when `/root/reference` is
absent (it does not exist on the GPU box).

The reference targets torch 1.0.0 (`/root/reference/.travis.yml:7`) and needs
three things that modern torch no longer has:

* `Tensor.btrifact(pivot=...)` / `Tensor.btrisolve(LU, piv)`  (used at
  `lcp_physics/lcp/solvers/pdipm.py:18,333,342,349,378,383,393,395`)
* uint8 masks accepted by `masked_scatter_` (`pdipm.py:423-428`)
* legacy instance-style `autograd.Function` (`lcp_physics/lcp/lcp.py:8-35`)

and `lcp_physics.physics` imports `ode` and `pygame`, for which
`oracle/_stubs/` provides import-only stand-ins.
"""
import os
import sys
import warnings

import torch

REFERENCE_ROOT = os.environ.get("LCP_REFERENCE_ROOT", "/root/reference")
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_stubs")
_installed = False


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lcp_physics"))


def _btrifact(self, pivot=True):
    # CPU torch only implements the pivoted factorisation; the reference asks for
    # pivot=not x.is_cuda (pdipm.py:18) which is True on CPU anyway.
    return torch.linalg.lu_factor(self, pivot=True)


def _btrisolve(self, LU, piv):
    if self.dim() == LU.dim() - 1:
        return torch.linalg.lu_solve(LU, piv, self.unsqueeze(-1)).squeeze(-1)
    return torch.linalg.lu_solve(LU, piv, self)


def install():
    """Patch torch and sys.path so that `import lcp_physics` resolves to the reference."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # never drop __pycache__ into /root/reference
    warnings.filterwarnings("ignore", category=UserWarning)

    torch.Tensor.btrifact = _btrifact
    torch.Tensor.btrisolve = _btrisolve

    orig_masked_scatter_ = torch.Tensor.masked_scatter_

    def masked_scatter_(self, mask, source):
        if mask.dtype == torch.uint8:
            mask = mask.bool()
        return orig_masked_scatter_(self, mask, source)

    torch.Tensor.masked_scatter_ = masked_scatter_

    for p in (_STUBS, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    _installed = True


class _RefLCP(torch.autograd.Function):
    """New-style adapter around the reference op.

    forward replays `lcp_physics/lcp/lcp.py:22-35`, backward `lcp.py:37-64`, by
    CALLING the reference's own `pdipm.pre_factor_kkt / forward / factor_kkt /
    solve_kkt` and `util.bger`; only the autograd plumbing is new.
    """

    @staticmethod
    def forward(ctx, Q, p, G, h, A, b, F, eps, verbose, not_improved_lim, max_iter, holder):
        from lcp_physics.lcp.solvers import pdipm
        _, nineq, nz = G.size()
        neq = A.size(1) if A.ndimension() > 1 else 0
        assert neq > 0 or nineq > 0
        Q_LU, S_LU, R = pdipm.pre_factor_kkt(Q, G, F, A)
        zhats, nus, lams, slacks = pdipm.forward(
            Q, p, G, h, A, b, F, Q_LU, S_LU, R, eps=eps, max_iter=max_iter,
            verbose=verbose, not_improved_lim=not_improved_lim)
        ctx.sizes = (neq, nineq, nz)
        ctx.kkt = (Q_LU, S_LU, R)
        ctx.duals = (nus, lams, slacks)
        ctx.save_for_backward(zhats, Q, p, G, h, A, b, F)
        if holder is not None:
            holder.nus, holder.lams, holder.slacks = nus, lams, slacks
            holder.Q_LU, holder.S_LU, holder.R = Q_LU, S_LU, R
            holder.neq, holder.nineq, holder.nz = neq, nineq, nz
        return zhats

    @staticmethod
    def backward(ctx, dl_dzhat):
        from lcp_physics.lcp.solvers import pdipm
        from lcp_physics.lcp.util import bger, extract_batch_size
        zhats, Q, p, G, h, A, b, F = ctx.saved_tensors
        neq, nineq, nz = ctx.sizes
        Q_LU, S_LU, R = ctx.kkt
        nus, lams, slacks = ctx.duals
        nb = extract_batch_size(Q, p, G, h, A, b)
        d = lams / slacks
        pdipm.factor_kkt(S_LU, R, d)
        dx, _, dlam, dnu = pdipm.solve_kkt(
            Q_LU, d, G, A, S_LU, dl_dzhat, G.new_zeros(nb, nineq),
            G.new_zeros(nb, nineq), G.new_zeros(nb, neq))
        dps = dx
        dGs = bger(dlam, zhats) + bger(lams, dx)
        dFs = -bger(dlam, lams)
        dhs = -dlam
        if neq > 0:
            dAs = bger(dnu, zhats) + bger(nus, dx)
            dbs = -dnu
        else:
            dAs, dbs = None, None
        dQs = 0.5 * (bger(dx, zhats) + bger(zhats, dx))
        return (dQs, dps, dGs, dhs, dAs, dbs, dFs, None, None, None, None, None)


class RefLCPFunction:
    """Call-compatible stand-in for the reference's legacy `LCPFunction` class."""

    def __init__(self, eps=1e-12, verbose=-1, not_improved_lim=3, max_iter=10):
        self.eps, self.verbose = eps, verbose
        self.not_improved_lim, self.max_iter = not_improved_lim, max_iter

    def __call__(self, Q, p, G, h, A, b, F):
        return _RefLCP.apply(Q, p, G, h, A, b, F, self.eps, self.verbose,
                             self.not_improved_lim, self.max_iter, self)


def load_reference():
    """Return the reference modules with the LCP op swapped for the adapter."""
    install()
    import lcp_physics.lcp.solvers.pdipm as pdipm
    import lcp_physics.lcp.util as lcp_util
    import lcp_physics.physics.engines as engines
    import lcp_physics.physics as physics
    engines.LCPFunction = RefLCPFunction
    return {"pdipm": pdipm, "util": lcp_util, "engines": engines, "physics": physics}

"""`LCPFunction`: the differentiable LCP op, MI355X-native.

Host-side mirror of the reference's `lcp_physics/lcp/lcp.py:8-64` - same constructor
keywords (`eps, verbose, not_improved_lim, max_iter`), same call signature
`instance(Q, p, G, h, A, b, F) -> zhats[B, nz]`, same attributes after the call
(`.nus, .lams, .slacks, .neq, .nineq, .nz`), same 7 gradients in input order, same error
for a singular Q (`lcp/solvers/pdipm.py:361-368`).  Modern torch rejects the reference's
legacy instance-style `autograd.Function`, so the class dispatches to a new-style Function.

All arithmetic happens in the HIP kernels behind the C ABI (`include/lcp_hip.h`); this file
only moves pointers.  There is no CPU fallback: without a GPU or without the built library
the call raises.
"""
import torch

from .. import _lib

SINGULAR_Q_MESSAGE = """
lcp Error: Cannot perform LU factorization on Q.
Please make sure that your Q matrix is PSD and has
a non-zero diagonal.
"""

_COMPUTE = {"f32": _lib.COMPUTE_F32, "f64": _lib.COMPUTE_F64, "fp32": _lib.COMPUTE_F32,
            "fp64": _lib.COMPUTE_F64, torch.float32: _lib.COMPUTE_F32, torch.float64: _lib.COMPUTE_F64}


def _pick_device(tensors):
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("lcp_physics_amd needs a GPU (MI355X); no CPU fallback exists")
    return torch.device("cuda", torch.cuda.current_device())


def _prep(t, B, ndim, device, dtype):
    """Batch-expand (lcp/util.py:49-55 `expandParam`), move and make contiguous."""
    if t is None or t.numel() == 0 and t.dim() <= 1:
        return None
    t = t.detach()
    if t.dim() == ndim - 1:
        t = t.unsqueeze(0).expand(B, *t.shape)
    return t.to(device=device, dtype=dtype).contiguous()


class LCPSolution:
    """Result of one batched solve; keeps the workspace the backward kernel needs."""
    __slots__ = ("x", "y", "z", "s", "iters", "status", "ws", "G", "A", "sizes", "compute", "dtype", "all_contact", "_bwd_plan", "adjoint_backward")


def lcp_solve(Q, p, G, h, A, b, F, eps=1e-12, not_improved_lim=3, max_iter=10, compute="f64",
              ws=None, out=None, path="auto"):
    """Forward solve on GPU tensors (no autograd).  Q,p,G,h,A,b,F: contiguous CUDA tensors of
    one dtype (float32 or float64), batched; A, b may be None.  Returns an `LCPSolution`.
    `path`: "auto" (the library's choice: contact-structured scenes with a diagonal Q are factored in BODY space - nz - neq or
    nz + neq rows), "big" (LCP_PATH_CONTACT_SPACE: the reference's own contact-space formulation, pdipm.py:357-454 - where a
    solve converges to rounding inside max_iter its exit tests then fall exactly where the reference's do), "generic"."""
    lib = _lib.load()
    dtype = G.dtype
    dev = G.device
    B, m, nz = G.shape
    e = A.shape[1] if A is not None else 0
    for name, t in (("Q", Q), ("p", p), ("G", G), ("h", h), ("F", F)):
        _lib.require_gpu_tensor(t, name, dtype)
    if e:
        _lib.require_gpu_tensor(A, "A", dtype)
        _lib.require_gpu_tensor(b, "b", dtype)
    assert Q.shape == (B, nz, nz) and p.shape == (B, nz) and h.shape == (B, m) and F.shape == (B, m, m)
    comp = (_lib.COMPUTE_F64 if dtype == torch.float64 else _COMPUTE[compute]) | _lib.path_bits(path)
    need = _lib.workspace_bytes(B, nz, m, e, comp | (_lib.IO_F64 if dtype == torch.float64 else 0))
    if need == 0:
        raise RuntimeError("invalid LCP sizes B=%d nz=%d nineq=%d neq=%d" % (B, nz, m, e))
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
    sol = out if out is not None else LCPSolution()
    if out is None:
        sol.x = torch.empty(B, nz, dtype=dtype, device=dev)
        sol.z = torch.empty(B, m, dtype=dtype, device=dev)
        sol.s = torch.empty(B, m, dtype=dtype, device=dev)
        sol.y = torch.empty(B, e, dtype=dtype, device=dev) if e else None
        sol.iters = torch.empty(B, dtype=torch.int32, device=dev)
        sol.status = torch.empty(B, dtype=torch.int32, device=dev)
    sol.ws, sol.G, sol.A, sol.sizes, sol.compute, sol.dtype = ws, G, A, (B, nz, m, e), comp, dtype
    P = _lib.ptr
    st = _lib.stream_ptr(dev)
    with torch.cuda.device(dev):
        if dtype == torch.float64:
            with _lib.thread_path(comp):
                rc = lib.lcp_pdipm_forward_f64(B, nz, m, e, P(Q), P(p), P(G), P(h), P(A), P(b), P(F),
                                               float(eps), int(max_iter), int(not_improved_lim),
                                               P(sol.x), P(sol.y), P(sol.z), P(sol.s), P(sol.iters),
                                               P(sol.status), P(ws), st)
        else:
            rc = lib.lcp_pdipm_forward_f32(B, nz, m, e, P(Q), P(p), P(G), P(h), P(A), P(b), P(F),
                                           float(eps), int(max_iter), int(not_improved_lim), comp,
                                           P(sol.x), P(sol.y), P(sol.z), P(sol.s), P(sol.iters),
                                           P(sol.status), P(ws), st)
    _lib.check(rc, "lcp_pdipm_forward")
    return sol


def _bwd_key(sol, dl_dx, out):
    return (dl_dx.data_ptr(), sol.G.data_ptr(), 0 if sol.A is None else sol.A.data_ptr(), sol.ws.data_ptr(), sol.compute,
            bool(getattr(sol, "all_contact", False)), tuple(0 if o is None else o.data_ptr() for o in out))


def lcp_backward(sol, dl_dx, need=(True,) * 7, out=None, adjoint=False):
    """Implicit-differentiation backward for a previous `lcp_solve` (lcp.py:37-64).
    `adjoint=True` (opt-in, never the parity target - SURVEY 0.5): solve with K^T instead of the reference's K - the exact gradient also where
    the LCP matrix F is not symmetric (LCP_BWD_ADJOINT; generic kernels: `sol` must come from `lcp_solve(..., path="generic")`).
    Returns [dQ, dp, dG, dh, dA, db, dF] (None where not needed / no equalities).  Called again with the `out` it returned and
    the same tensors it re-uses the validated argument list (one ctypes call; keyed on every pointer it holds).  With `out`
    given, `out` decides which gradients are written (its None entries are skipped); `need` only shapes a NEW `out`."""
    lib = _lib.load()
    B, nz, m, e = sol.sizes
    dev, dtype = sol.G.device, sol.dtype
    plan = getattr(sol, "_bwd_plan", None)
    # the cached argument list holds raw pointers: it is valid only while EVERY tensor behind it is the same - the cotangent, G, A,
    # the workspace and each gradient tensor of `out` (an entry the caller replaced gets a fresh list, not a write to the old tensor)
    if adjoint and not (sol.compute & _lib.PATH_GENERIC):
        raise ValueError("lcp_backward(adjoint=True) needs a solution of lcp_solve(..., path='generic')")
    if (plan is not None and out is not None and plan[0] is out and dl_dx.dtype == dtype and dl_dx.is_contiguous()
            and plan[1] == _bwd_key(sol, dl_dx, out) and not adjoint):
        from ..physics.batched_world import _on_device
        with _on_device(dev):
            rc = plan[2](*plan[3], _lib.stream_ptr(dev))
        _lib.check(rc, "lcp_pdipm_backward")
        return out
    dl_dx = _lib.require_gpu_tensor(dl_dx.to(dtype).contiguous(), "dl_dx", dtype)
    shapes = [(B, nz, nz), (B, nz), (B, m, nz), (B, m), (B, e, nz), (B, e), (B, m, m)]
    if out is None:
        out = [torch.empty(sh, dtype=dtype, device=dev) if (nd and (e or i not in (4, 5))) else None
               for i, (sh, nd) in enumerate(zip(shapes, need))]
    P = _lib.ptr
    st = _lib.stream_ptr(dev)
    with torch.cuda.device(dev):
        if dtype == torch.float64:
            with _lib.thread_path(sol.compute):
                lib.lcp_set_backward_adjoint(1 if adjoint else 0)
                try:
                    rc = lib.lcp_pdipm_backward_f64(B, nz, m, e, P(sol.G), P(sol.A), P(dl_dx),
                                                    *[P(o) for o in out], P(sol.ws), st)
                finally:
                    lib.lcp_set_backward_adjoint(0)
        else:
            hint = _lib.HINT_ALL_CONTACT if getattr(sol, "all_contact", False) else 0
            args = (B, nz, m, e, P(sol.G), P(sol.A), P(dl_dx), sol.compute | hint | (_lib.BWD_ADJOINT if adjoint else 0), *[P(o) for o in out], P(sol.ws))
            rc = lib.lcp_pdipm_backward_f32(*args, st)
            if not adjoint:
                sol._bwd_plan = (out, _bwd_key(sol, dl_dx, out), lib.lcp_pdipm_backward_f32, args)
    _lib.check(rc, "lcp_pdipm_backward")
    return out


class _LCPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, p, G, h, A, b, F, holder):
        ins = (Q, p, G, h, A, b, F)
        if G.dim() != 3:
            raise RuntimeError("G must be [batch, nineq, nz] (lcp/lcp.py:23)")
        B, nineq, nz = G.shape
        neq = A.size(1) if A.dim() > 1 else 0                      # lcp.py:24
        assert neq > 0 or nineq > 0                                  # lcp.py:25
        dev = _pick_device(ins)
        dtype = G.dtype if G.dtype in (torch.float32, torch.float64) else torch.float32
        dims = (3, 2, 3, 2, 3, 2, 3)
        dQ, dp_, dG, dh_, dA, db_, dF = [_prep(t, B, nd, dev, dtype) for t, nd in zip(ins, dims)]
        if neq == 0:
            dA, db_ = None, None
        sol = lcp_solve(dQ, dp_, dG, dh_, dA, db_, dF, eps=holder.eps,
                        not_improved_lim=holder.not_improved_lim, max_iter=holder.max_iter,
                        compute=holder.compute, path=holder.path)
        if holder.check:
            st = sol.status.cpu()
            if bool((st & _lib.ST_SINGULAR_Q).any()):
                raise RuntimeError(SINGULAR_Q_MESSAGE)                # pdipm.py:361-368
        back = lambda t: None if t is None else t.to(device=G.device, dtype=G.dtype)
        holder.nus, holder.lams, holder.slacks = back(sol.y), back(sol.z), back(sol.s)   # lcp.py:29
        holder.neq, holder.nineq, holder.nz = neq, nineq, nz
        holder.iters, holder.status, holder.solution = sol.iters, sol.status, sol
        # the handle the backward takes, WITHOUT x: `back(sol.x)` below is sol.x itself when no conversion is needed, autograd stamps
        # this node on it, and a ctx that holds it would be a cycle the garbage collector cannot see (the workspace would never be freed)
        import copy
        ctx.sol = copy.copy(sol)
        ctx.sol.x = None
        ctx.sol.adjoint_backward = holder.adjoint_backward
        ctx.meta = [(t.device, t.dtype, tuple(t.shape)) for t in ins]
        ctx.neq = neq
        return back(sol.x)

    @staticmethod
    def backward(ctx, dl_dzhat):
        sol = ctx.sol
        need = list(ctx.needs_input_grad[:7])
        g = dl_dzhat.to(device=sol.G.device, dtype=sol.dtype).contiguous()
        grads = lcp_backward(sol, g, need=need, adjoint=getattr(sol, "adjoint_backward", False))
        out = []
        for i, (gr, (dev, dt, shape)) in enumerate(zip(grads, ctx.meta)):
            if gr is None or not need[i]:
                out.append(None)
                continue
            gr = gr.to(device=dev, dtype=dt)
            if gr.dim() == len(shape) + 1:                            # un-batched parameter: sum over batch
                gr = gr.sum(0)
            out.append(gr)
        return tuple(out) + (None,)


class LCPFunction:
    """A differentiable LCP solver (primal-dual interior point), drop-in for the reference class.

    Extra keywords (not in the reference): `compute` - arithmetic used inside the kernels for
    float32 inputs ("f64" = parity path, "f32" = all-fp32 fast path); `check` - read the status
    word back (one host sync) and raise on a singular Q like the reference does; `path` - kernel
    family, see `lcp_solve` ("big" = the reference's contact-space formulation instead of the default
    body-space factorisation of contact-structured scenes).
    """

    def __init__(self, eps=1e-12, verbose=-1, not_improved_lim=3, max_iter=10, compute="f64", check=True, path="auto",
                 adjoint_backward=False):
        # adjoint_backward (not in the reference, opt-in): the backward solves with K^T - the exact gradient where F is not symmetric; the
        # reference's own formula (lcp.py:46-50, K itself) stays the default and the parity target.  Generic kernels (LCP_PATH_GENERIC).
        self.adjoint_backward = bool(adjoint_backward)
        self.path = "generic" if adjoint_backward else path
        self.eps = eps
        self.verbose = verbose
        self.not_improved_lim = not_improved_lim
        self.max_iter = max_iter
        self.compute = compute
        self.check = check
        self.Q_LU = self.S_LU = self.R = None       # kept on the device workspace instead
        self.nus = self.lams = self.slacks = None

    def __call__(self, Q, p, G, h, A, b, F):
        return _LCPFn.apply(Q, p, G, h, A, b, F, self)

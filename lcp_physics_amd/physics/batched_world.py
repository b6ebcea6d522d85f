"""Batched scene container and the fused simulation step.

The reference's physics layer is un-batched (`World` steps ONE scene and calls the solver with
batch 1, `physics/engines.py:54-76`).  `BatchedWorld` is the new batched surface:
B independent scenes stored structure-of-arrays in HBM (`scenes.SceneBatch`), advanced by one
HIP launch per step - contact-Jacobian assembly (`world.py:144-234`), `u = M v + dt f`
(`engines.py:31-32`), the PDIPM LCP solve (`lcp/solvers/pdipm.py`), `new_v = -x`
(`engines.py:76-77`) and the semi-implicit integrator `p += v dt` (`bodies.py:80-82`).
Its `step()` has the semantics of the reference's `World.step_dt` for a fixed contact list;
contact detection (`physics/contacts.py`) is SURVEY.md §8(f) row 1 ("next") and is supplied by
the caller through `contact_fn` until then.
"""
import torch

from .. import _lib
from ..lcp.lcp import _COMPUTE, LCPSolution
from ..scenes import SceneBatch


def scene_from_contacts(p, v, Mdiag, f, rest, fric, contacts, Je, dt):
    """Build a 1-scene `SceneBatch` from a reference-style contact list
    `[((normal, p1, p2, penetration), i1, i2), ...]` (`physics/contacts.py:203-204`)."""
    st = lambda k: torch.stack([c[0][k].detach().reshape(2) for c in contacts]).unsqueeze(0)
    idx = lambda k: torch.tensor([[int(c[k]) for c in contacts]], dtype=torch.int32)
    nb = v.shape[0]
    u = lambda t: t.detach().unsqueeze(0)
    Je_b = Je.detach().unsqueeze(0) if Je is not None else torch.zeros(1, 0, 3 * nb, dtype=v.dtype)
    return SceneBatch(p=u(p), v=u(v), Mdiag=u(Mdiag), f=u(f), rest=u(rest), fric=u(fric),
                      c_n=st(0), c_p1=st(1), c_p2=st(2), c_i1=idx(1), c_i2=idx(2), Je=Je_b, dt=dt)


def _check_scene(sc):
    for name in ("p", "v", "Mdiag", "f", "rest", "fric", "c_n", "c_p1", "c_p2"):
        _lib.require_gpu_tensor(getattr(sc, name), name, torch.float32)
    _lib.require_gpu_tensor(sc.c_i1, "c_i1", torch.int32)
    _lib.require_gpu_tensor(sc.c_i2, "c_i2", torch.int32)
    e = sc.Je.shape[1] if sc.Je is not None and sc.Je.numel() else 0
    if e:
        _lib.require_gpu_tensor(sc.Je, "Je", torch.float32)
    return e


def assemble_contacts(sc):
    """Dense (Q, p, G, h, A, b, F) of `engines.py:50-74` for every scene, built by the HIP
    assembly kernel.  Returns float32 CUDA tensors (A, b are None without joints)."""
    lib = _lib.load()
    e = _check_scene(sc)
    B, nb, nc = sc.B, sc.nb, sc.nc
    nz, m = 3 * nb, 4 * nc
    dev = sc.v.device
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    Q, p, G, h, F = new(B, nz, nz), new(B, nz), new(B, m, nz), new(B, m), new(B, m, m)
    A, b = (new(B, e, nz), new(B, e)) if e else (None, None)
    P = _lib.ptr
    with _on_device(dev):
        rc = lib.lcp_assemble_contacts_f32(B, nb, nc, e, P(sc.Mdiag), P(sc.v), P(sc.f), P(sc.rest),
                                           P(sc.fric), P(sc.c_n), P(sc.c_p1), P(sc.c_p2), P(sc.c_i1),
                                           P(sc.c_i2), P(sc.Je) if e else None, float(sc.dt),
                                           P(Q), P(p), P(G), P(h), P(A), P(b), P(F), _lib.stream_ptr(dev))
    _lib.check(rc, "lcp_assemble_contacts_f32")
    return Q, p, G, h, A, b, F


def rows_pin_leading_coordinates(Je):
    """Host check (synchronises once): every scene's equality rows are Je = [I 0] - the TotalConstraint that fixes the floor of
    the reference's demo worlds (`constraints.py:175-192`) -, or there are none.  What `LCP_HINT_PINNED` promises."""
    if Je is None or Je.numel() == 0:
        return True
    e, nz = Je.shape[1], Je.shape[2]
    if e > nz:
        return False
    eye = torch.zeros(e, nz, dtype=Je.dtype, device=Je.device)
    eye[:, :e] = torch.eye(e, dtype=Je.dtype, device=Je.device)
    return bool((Je == eye).all())


def _pinned_hint(sc, pinned):
    """`pinned=None`: decided once per SceneBatch from its Je (cached on the object; Je must not be edited in place afterwards)."""
    if pinned is None:
        pinned = getattr(sc, "_rows_pinned", None)
        if pinned is None:
            pinned = rows_pin_leading_coordinates(sc.Je)
            try:
                sc._rows_pinned = pinned
            except AttributeError:
                pass
    return _lib.HINT_PINNED if pinned else 0


def _on_device(dev):
    """`with torch.cuda.device(dev)` costs several microseconds per call: entered only when `dev` is not already current."""
    import contextlib
    if dev.index is None or torch.cuda.current_device() == dev.index:
        return contextlib.nullcontext()
    return torch.cuda.device(dev)


_SCENE_FIELDS = ("p", "Mdiag", "v", "f", "rest", "fric", "c_n", "c_p1", "c_p2", "c_i1", "c_i2")
_STEP_OUTPUTS = ("v_new", "p_new", "z", "s", "y", "iters", "status")
_STEP_KEY_LEN = 10                       # entries of fused_step's plan key before the output pointers


def fused_step(sc, eps=1e-12, not_improved_lim=3, max_iter=10, compute="f64", ws=None, out=None, path="auto", pinned=None,
               multipliers=True):
    """One fused simulation step for every scene of `sc` (float32 CUDA `SceneBatch`).

    Returns a dict with v_new, p_new [B,nb,3], z, s [B,4nc], y [B,e], iters, status [B] and the
    workspace `ws` (re-usable; it also feeds `lcp_backward`).  `pinned`: the equality rows of every scene pin the leading
    coordinates (`LCP_HINT_PINNED`: one launch less); None = checked once per SceneBatch on the host.  `multipliers=False`: z, s (and
    y) are not written out - the reference's step returns new_v only (`engines.py:76-77`; its multipliers stay inside the op for the
    backward, here: in fp64 in the workspace) - which saves the step a third of its HBM writes.
    Calling it again with the `out` / `ws` it returned and the same tensors re-uses the validated argument list (the host side of
    a step is then one ctypes call: at small batches the step is otherwise bound by this wrapper, not by the GPU); the list is
    keyed on every pointer it holds - scene, outputs, workspace - and on the options.  With `out` given, `out` decides what is
    written (its `z` / `s` / `y` entries, tensors or None); `multipliers` only shapes a NEW `out`."""
    lib = _lib.load()
    dev = sc.v.device
    key = (tuple(getattr(sc, k).data_ptr() for k in _SCENE_FIELDS), 0 if sc.Je is None else sc.Je.data_ptr(), float(sc.dt),
           float(eps), int(not_improved_lim), int(max_iter), compute, path, pinned, _lib.path_bits(path))
    if out is not None:
        # ... and on the OUTPUT tensors the caller hands back: an entry of `out` that was replaced (to keep the old result) or a
        # different workspace gets a fresh argument list, never a write through a stale pointer
        key += (tuple(0 if out.get(k) is None else out[k].data_ptr() for k in _STEP_OUTPUTS),
                0 if out.get("ws") is None else out["ws"].data_ptr())
    plan = None if out is None else out.get("_plan")
    if plan is not None and plan[0] == key and (ws is None or ws is out["ws"]):
        with _on_device(dev):
            rc = lib.lcp_step_fused_f32(*plan[1], _lib.stream_ptr(dev))
        _lib.check(rc, "lcp_step_fused_f32")
        return out
    e = _check_scene(sc)
    B, nb, nc = sc.B, sc.nb, sc.nc
    nz, m = 3 * nb, 4 * nc
    comp = _COMPUTE[compute]
    need = _lib.workspace_bytes(B, nz, m, e, comp)
    comp |= _lib.path_bits(path) | _pinned_hint(sc, pinned)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
    if out is None:
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        out = {"v_new": new(B, nb, 3), "p_new": new(B, nb, 3), "z": new(B, m) if multipliers else None,
               "s": new(B, m) if multipliers else None, "y": new(B, e) if (e and multipliers) else None,
               "iters": torch.empty(B, dtype=torch.int32, device=dev),
               "status": torch.empty(B, dtype=torch.int32, device=dev)}
    out["ws"] = ws
    P = _lib.ptr
    args = (B, nb, nc, e, P(sc.p), P(sc.Mdiag), P(sc.v), P(sc.f), P(sc.rest), P(sc.fric), P(sc.c_n), P(sc.c_p1), P(sc.c_p2),
            P(sc.c_i1), P(sc.c_i2), P(sc.Je) if e else None, float(sc.dt), float(eps), int(max_iter), int(not_improved_lim), comp,
            P(out["v_new"]), P(out["p_new"]), P(out["z"]), P(out["s"]), P(out["y"]), P(out["iters"]), P(out["status"]), P(ws))
    with _on_device(dev):
        rc = lib.lcp_step_fused_f32(*args, _lib.stream_ptr(dev))
    _lib.check(rc, "lcp_step_fused_f32")
    out["compute"] = comp                 # the word the backward must carry: arithmetic + kernel path (a family per word, any thread)
    key = key[:_STEP_KEY_LEN] + (tuple(0 if out.get(k) is None else out[k].data_ptr() for k in _STEP_OUTPUTS), ws.data_ptr())
    out["_plan"] = (key, args)            # (valid while the scene, the options, the output tensors and the workspace are the same)
    return out


def _word(out, compute):
    """`compute` word of the forward that produced `out` (recorded there); a handle from before falls back on the arithmetic."""
    return out["compute"] if "compute" in out else (_COMPUTE[compute] | _lib.path_bits())


def fused_step_backward(sc, out, dl_dv, compute="f64", grads=None, want_Je=False):
    """Backward of `fused_step` with respect to the physical inputs of the scenes: d(loss)/d(v_new) [B,nb,3] ->
    dict(Mdiag, v, f [B,nb,3], rest, fric [B,nb], c_n, c_p1, c_p2 [B,nc,2]) - what the reference computes by
    autograd through `engines.py:31-32,50-77` and `world.py:144-234` after `LCPFunction.backward`.  One launch of
    `lcp_step_backward_je_f32`; the dense LCP gradients are never materialised.  `out` is the dict `fused_step`
    returned (its workspace is read); the scene must not have changed in between.  `want_Je`: also "Je" [B,e,3nb], the
    gradient of the joint Jacobian (`lcp.py:57`, dA = dnu x^T + nu dx^T)."""
    lib = _lib.load()
    from .dense_step import require_fused_record
    require_fused_record(out, "fused_step_backward")
    e = _check_scene(sc)
    B, nb, nc = sc.B, sc.nb, sc.nc
    dev = sc.v.device
    dl_dv = _lib.require_gpu_tensor(dl_dv.to(torch.float32).contiguous(), "dl_dv", torch.float32)
    if grads is None:
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        grads = {"Mdiag": new(B, nb, 3), "v": new(B, nb, 3), "f": new(B, nb, 3), "rest": new(B, nb), "fric": new(B, nb),
                 "c_n": new(B, nc, 2), "c_p1": new(B, nc, 2), "c_p2": new(B, nc, 2)}
    if want_Je and e and "Je" not in grads:
        grads["Je"] = torch.empty(B, e, 3 * nb, dtype=torch.float32, device=dev)
    P = _lib.ptr
    with _on_device(dev):
        rc = lib.lcp_step_backward_je_f32(B, nb, nc, e, P(sc.Mdiag), P(sc.v), P(sc.f), P(sc.rest), P(sc.fric), P(sc.c_n),
                                          P(sc.c_p1), P(sc.c_p2), P(sc.c_i1), P(sc.c_i2), P(sc.Je) if e else None,
                                          float(sc.dt), P(dl_dv), _word(out, compute),
                                          P(grads["Mdiag"]), P(grads["v"]),
                                          P(grads["f"]), P(grads["rest"]), P(grads["fric"]), P(grads["c_n"]),
                                          P(grads["c_p1"]), P(grads["c_p2"]), P(grads["Je"]) if (want_Je and e) else None,
                                          P(out["ws"]), _lib.stream_ptr(dev))
    _lib.check(rc, "lcp_step_backward_je_f32")
    return grads


def solution_of_step(sc, out, G, A, compute="f64"):
    """Wrap a fused step's workspace as an `LCPSolution` so `lcp.lcp_backward` can follow
    (G, A from `assemble_contacts`)."""
    from .dense_step import require_fused_record
    require_fused_record(out, "solution_of_step")
    sol = LCPSolution()
    B, nb, nc = sc.B, sc.nb, sc.nc
    e = A.shape[1] if A is not None else 0
    sol.x = None                                   # x = -v_new (engines.py:76-77); the backward reads it from the workspace
    sol.all_contact = True                         # every scene went through the contact-structured kernel
    sol.y, sol.z, sol.s = out["y"], out["z"], out["s"]
    sol.iters, sol.status, sol.ws = out["iters"], out["status"], out["ws"]
    sol.G, sol.A, sol.sizes, sol.dtype = G, A, (B, 3 * nb, 4 * nc, e), torch.float32
    sol.compute = _word(out, compute)              # the backward must pick the forward's kernel family and workspace layout
    return sol


class BatchedWorld:
    """B independent scenes advanced together on one GPU.

    world = BatchedWorld(scene_batch.to("cuda"), contact_fn=None)
    world.step()            # one fused HIP launch; updates world.scene.p / .v, world.t

    `contact_fn(world) -> (c_n, c_p1, c_p2, c_i1, c_i2)` may refresh the contact list after the
    integrator moved the bodies (the role of `World.find_contacts`, `world.py:139-142`)."""

    def __init__(self, scene, contact_fn=None, max_iter=10, compute="f64", eps=1e-12, not_improved_lim=3):
        self.scene = scene
        self.contact_fn = contact_fn
        self.max_iter, self.compute, self.eps, self.lim = max_iter, compute, eps, not_improved_lim
        self.t = 0.0
        self.dt = scene.dt
        self._ws = None
        self._out = None
        self._count = None
        self.last = None

    def step_autograd(self):
        """The same step as a node of torch's autograd graph (`demos/grad_demo.py:45-50`, `experiments/inference.py:55-61`
        back-propagate through many `World.step`s): `SolveDynamicsFunction` (HIP forward + HIP analytic backward) for the
        velocities and `p + v_new dt` (`bodies.py:80-82`) for the poses.  Scene tensors may require grad (masses, forces,
        restitution, friction, initial velocities, the contact frame); the state tensors are REPLACED, not overwritten, so
        every step of a roll-out keeps what its backward needs."""
        from dataclasses import replace
        sc = self.scene
        if self._count is None or self._count.shape[0] != sc.B:
            self._count = torch.full((sc.B,), sc.nc, dtype=torch.int32, device=sc.v.device)
        opts = {"max_iter": self.max_iter, "eps": self.eps, "not_improved_lim": self.lim, "compute": self.compute}
        e = sc.Je.shape[1] if sc.Je is not None and sc.Je.numel() else 0
        v_new = SolveDynamicsFunction.apply(sc.Mdiag, sc.v, sc.f, sc.rest, sc.fric, sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1,
                                            sc.c_i2, self._count, sc.Je if e else None, sc.dt, opts)
        p_new = sc.p + v_new * sc.dt
        ret = dict(opts["last"])
        ret["v_prev"], ret["p_prev"], ret["v_new"], ret["p_new"] = sc.v, sc.p, v_new, p_new
        self.scene = replace(sc, v=v_new, p=p_new)
        self.last = ret
        if self.contact_fn is not None:
            c_n, c_p1, c_p2, c_i1, c_i2 = self.contact_fn(self)
            self.scene = replace(self.scene, c_n=c_n, c_p1=c_p1, c_p2=c_p2, c_i1=c_i1, c_i2=c_i2)
        self.t += self.dt
        return ret

    def step(self, differentiable=False):
        if differentiable:
            return self.step_autograd()
        sc = self.scene
        # (lcp_step_fused_f32 picks the kernel family from the sizes: four scenes per wave up to 16 contacts / 10 bodies,
        # the register-tiled workgroup kernel up to 64 contacts, the generic kernels beyond)
        out = fused_step(sc, eps=self.eps, not_improved_lim=self.lim, max_iter=self.max_iter,
                         compute=self.compute, ws=self._ws, out=self._out)
        self._ws, self._out = out["ws"], out
        # double-buffer swap: new state becomes the scene state (world.py:87-90); the dict handed back (and kept as
        # `last`) names the NEW tensors, the spare buffers stay private
        self.scene.v, out["v_new"] = out["v_new"], self.scene.v
        self.scene.p, out["p_new"] = out["p_new"], self.scene.p
        ret = dict(out)
        ret["v_new"], ret["p_new"] = self.scene.v, self.scene.p
        ret["v_prev"], ret["p_prev"] = out["v_new"], out["p_new"]          # state the step started from (until the next step)
        self.last = ret
        if self.contact_fn is not None:
            c_n, c_p1, c_p2, c_i1, c_i2 = self.contact_fn(self)
            self.scene.c_n, self.scene.c_p1, self.scene.c_p2 = c_n, c_p1, c_p2
            self.scene.c_i1, self.scene.c_i2 = c_i1, c_i2
        self.t += self.dt
        return ret

    def get_v(self):
        return self.scene.v

    def get_p(self):
        return self.scene.p


def solve_dynamics(B, nb, maxc, e, count, Mdiag, v, f, rest, fric, cb, Je, dt, eps=1e-12, not_improved_lim=3,
                   max_iter=10, compute="f64", ws=None, out=None, path="auto", pinned=False):
    """`PdipmEngine.solve_dynamics` (`engines.py:26-78`) for B scenes with per-scene contact counts
    (`count` [B] int32, contact records in `cb`, a `contacts.ContactBuffers`): one launch of
    `lcp_solve_dynamics_f32`.  Returns dict(v_new, z, s, y, iters, status, ws)."""
    lib = _lib.load()
    dev = v.device
    comp = _COMPUTE[compute]
    nz, m = 3 * nb, 4 * maxc
    need = _lib.workspace_bytes(B, nz, m, e, comp)
    comp |= _lib.path_bits(path) | (_lib.HINT_PINNED if pinned else 0)      # (pinned: LCP_HINT_PINNED, the caller checked its Je)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
    if out is None:
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        out = {"v_new": new(B, nb, 3), "z": new(B, m), "s": new(B, m), "y": new(B, e) if e else None,
               "iters": torch.empty(B, dtype=torch.int32, device=dev),
               "status": torch.empty(B, dtype=torch.int32, device=dev)}
    out["ws"] = ws
    P = _lib.ptr
    with _on_device(dev):
        rc = lib.lcp_solve_dynamics_f32(B, nb, maxc, e, P(count), P(Mdiag), P(v), P(f), P(rest), P(fric),
                                        P(cb.c_n), P(cb.c_p1), P(cb.c_p2), P(cb.c_i1), P(cb.c_i2),
                                        P(Je) if e else None, float(dt), float(eps), int(max_iter),
                                        int(not_improved_lim), comp, P(out["v_new"]), P(out["z"]), P(out["s"]),
                                        P(out["y"]), P(out["iters"]), P(out["status"]), P(ws), _lib.stream_ptr(dev))
    _lib.check(rc, "lcp_solve_dynamics_f32")
    out["compute"] = comp
    return out


def solve_dynamics_backward(B, nb, maxc, e, Mdiag, v, f, rest, fric, cb, Je, dt, dl_dv, out, compute="f64", grads=None,
                            want_Je=False):
    """Backward of `solve_dynamics` with respect to its physical inputs (what the reference gets by autograd through
    `engines.py:31-32,50-77`, `world.py:144-234` and `lcp.py:37-64`): one launch of `lcp_step_backward_je_f32` on the
    workspace the forward left in `out`.  Padded contact slots get zero gradients.
    Returns dict(Mdiag, v, f [B,nb,3], rest, fric [B,nb], c_n, c_p1, c_p2 [B,maxc,2]) and, with `want_Je`, Je [B,e,3nb] - the
    gradient of the joint Jacobian (`lcp.py:57` with A = Je), which worlds with pose-dependent joints propagate on."""
    lib = _lib.load()
    from .dense_step import require_fused_record
    require_fused_record(out, "solve_dynamics_backward")
    dev = v.device
    dl_dv = _lib.require_gpu_tensor(dl_dv.to(torch.float32).contiguous(), "dl_dv", torch.float32)
    if grads is None:
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        grads = {"Mdiag": new(B, nb, 3), "v": new(B, nb, 3), "f": new(B, nb, 3), "rest": new(B, nb), "fric": new(B, nb),
                 "c_n": new(B, maxc, 2), "c_p1": new(B, maxc, 2), "c_p2": new(B, maxc, 2)}
    if want_Je and e and "Je" not in grads:
        grads["Je"] = torch.empty(B, e, 3 * nb, dtype=torch.float32, device=dev)
    P = _lib.ptr
    with _on_device(dev):
        rc = lib.lcp_step_backward_je_f32(B, nb, maxc, e, P(Mdiag), P(v), P(f), P(rest), P(fric), P(cb.c_n), P(cb.c_p1),
                                          P(cb.c_p2), P(cb.c_i1), P(cb.c_i2), P(Je) if e else None, float(dt), P(dl_dv),
                                          _word(out, compute),
                                          P(grads["Mdiag"]), P(grads["v"]), P(grads["f"]), P(grads["rest"]), P(grads["fric"]),
                                          P(grads["c_n"]), P(grads["c_p1"]), P(grads["c_p2"]),
                                          P(grads["Je"]) if (want_Je and e) else None, P(out["ws"]), _lib.stream_ptr(dev))
    _lib.check(rc, "lcp_step_backward_je_f32")
    return grads


class _ValueOf(torch.autograd.Function):
    """y = `value` (bitwise: a copy of what the kernel computed) with the gradient of `lin`, the torch expression of the same
    quantity.  `lin + (value - lin).detach()` can differ from `value` by an ulp - and the contact list of the next step was
    detected at exactly `value` (a borderline pair must not change sides between the forward and the frame backward)."""

    @staticmethod
    def forward(ctx, lin, value):
        return value.detach().clone()                 # (`value` may be a buffer the next launch overwrites)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _Frame:
    """The contact list fields the entry points read (a `contacts.ContactBuffers` also qualifies)."""
    __slots__ = ("c_n", "c_p1", "c_p2", "c_i1", "c_i2")

    def __init__(self, c_n, c_p1, c_p2, c_i1, c_i2):
        self.c_n, self.c_p1, self.c_p2, self.c_i1, self.c_i2 = c_n, c_p1, c_p2, c_i1, c_i2


class _StateUpdate(torch.autograd.Function):
    """The state a differentiable step ends with, as ONE node:

        p_new, geo_new[, rot_new] = _StateUpdate.apply(p, p_geo, rot, v, p_out, rot_value, dt_used, scale, joints)

    values: `p_out`, the pose the kernels accepted (bitwise: the next contact list was detected at exactly these values), twice, and
    `rot_value`, the revolute joints' angles after the move; gradients: those of

        p_new   = p + scale v dt_used                                                          (bodies.py:80-82)
        geo_new = p_geo + [scale v dt_used where the increment is non-zero or the coordinate is x / y]   (bodies.py:199-202)
        rot_new = rot + scale v[body1][0] dt_used  for revolute joints                         (constraints.py:39-43)

    Nothing is launched in the forward pass (the sums are never evaluated, only differentiated); the backward is one launch of
    `lcp_state_update_backward_f64`.  `p_out`, `rot_value` and `dt_used` belong to the step (a differentiable step retires its contact
    buffers instead of re-using them).  `rot` / `rot_value` / `joints` are None for worlds without pose-dependent joints."""

    @staticmethod
    def forward(ctx, p, p_geo, rot, v, p_out, rot_value, dt_used, scale, joints):
        ctx.save_for_backward(v, dt_used)
        ctx.scale, ctx.joints = float(scale), joints
        if rot is None:
            return p_out.detach(), p_out.detach()
        return p_out.detach(), p_out.detach(), rot_value.detach()

    @staticmethod
    def backward(ctx, g_p, g_g, g_rot=None):
        v, dt_used = ctx.saved_tensors
        lib = _lib.load()
        B, nb = v.shape[0], v.shape[1]
        js = ctx.joints
        c64 = lambda g: None if g is None else g.to(torch.float64).contiguous()
        g_p, g_g, g_rot = c64(g_p), c64(g_g), (c64(g_rot) if js is not None else None)
        g_v = torch.empty(B, nb, 3, dtype=torch.float32, device=v.device)
        P = _lib.ptr
        with _on_device(v.device):
            rc = lib.lcp_state_update_backward_f64(B, nb, 0 if g_rot is None else js.jtype.shape[1], P(g_p), P(g_g), P(g_rot), P(v), P(dt_used),
                                                   ctx.scale, P(js.jtype) if g_rot is not None else None,
                                                   P(js.jb1) if g_rot is not None else None, P(g_v), _lib.stream_ptr(v.device))
        _lib.check(rc, "lcp_state_update_backward_f64")
        return g_p, g_g, g_rot, g_v, None, None, None, None, None


class _JointJacobianFn(torch.autograd.Function):
    """Je = the joint Jacobian the kernel computed at pose `p` and revolute angles `rot` (`JointSet.jacobian`; its values), with the
    gradient of `Joint.J()` / `FixedJoint.J()` (constraints.py:26-50, 64-85): backward = one launch of
    `lcp_joint_jacobian_backward_f64`.  `Je_value` belongs to the step (a fresh tensor per `jacobian()` call)."""

    @staticmethod
    def forward(ctx, p, rot, r1, joints, Je_value):
        ctx.joints, ctx.nb = joints, p.shape[1]
        ctx.save_for_backward(rot)
        return Je_value.detach()

    @staticmethod
    def backward(ctx, gJe):
        (rot,) = ctx.saved_tensors
        g_p, g_rot = ctx.joints.jacobian_backward(ctx.nb, rot, gJe)
        # (r1 = the anchors' radii: constants unless the joints were created at poses that require grad, constraints.py:21-23)
        g_r1 = ctx.joints.anchor_radius_backward(ctx.nb, rot, gJe) if ctx.needs_input_grad[2] else None
        return g_p, g_rot, g_r1, None, None


class SolveDynamicsFunction(torch.autograd.Function):
    """`PdipmEngine.solve_dynamics` (`engines.py:26-78`) for B scenes as ONE differentiable op.

        v_new = SolveDynamicsFunction.apply(Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2, count, Je, dt, opts)

    forward  = `lcp_solve_dynamics_f32` (assembly + PDIPM solve + `new_v = -x` in one launch; a scene with count 0
               takes the direct KKT solve of `engines.py:36-50`);
    backward = `lcp_step_backward_f32` (implicit differentiation, `lcp.py:37-64`, contracted through the assembly):
               gradients for Mdiag, v, f, rest, fric and the contact frame (c_n, c_p1, c_p2).
    All tensors float32, contiguous, on the GPU ([B,nb,3], [B,nb], [B,maxc,2], int32 [B,maxc] / [B]); `Je` [B,e,3nb] or
    None (its gradient - `lcp.py:57` with A = Je - is returned when the caller's Je requires one: worlds whose joints follow
    the pose); `opts`: dict(max_iter, eps, not_improved_lim, compute) - it receives the forward's
    `out` dict under "last" (z, s, y, iters, status).  Every call owns its workspace, so the steps of a roll-out can be
    back-propagated in reverse order."""

    @classmethod
    def apply(cls, Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2, count, Je, dt, opts):
        # The generic kernels step any size forward but keep no iterate a fused backward could read.  A step that is being RECORDED at
        # such a size goes through the dense boundary instead, the way the reference does at every size (physics/dense_step.py):
        # decided here, when the step is recorded - not with LCP_E_TOOLARGE in the middle of loss.backward().  (Inside forward() grad
        # mode is off.)
        if count is None:                                                  # (every scene uses all of its contact slots)
            count = torch.full((v.shape[0],), c_n.shape[1], dtype=torch.int32, device=v.device)
        args = (Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2, count, Je, dt, opts)
        if torch.is_grad_enabled() and any(isinstance(a, torch.Tensor) and a.requires_grad for a in args):
            nb, maxc = v.shape[1], c_n.shape[1]
            e = 0 if Je is None or Je.numel() == 0 else Je.shape[1]
            if not _lib.load().lcp_step_has_backward(nb, maxc, e, _COMPUTE[opts.get("compute", "f64")] | _lib.path_bits()):
                from .dense_step import solve_dynamics_dense
                return solve_dynamics_dense(*args)
        return super(SolveDynamicsFunction, cls).apply(*args)

    @staticmethod
    def forward(ctx, Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2, count, Je, dt, opts):
        B, nb = v.shape[0], v.shape[1]
        maxc = c_n.shape[1]
        e = 0 if Je is None or Je.numel() == 0 else Je.shape[1]
        for name, t in (("Mdiag", Mdiag), ("v", v), ("f", f), ("rest", rest), ("fric", fric), ("c_n", c_n),
                        ("c_p1", c_p1), ("c_p2", c_p2)):
            _lib.require_gpu_tensor(t, name, torch.float32)
        for name, t in (("c_i1", c_i1), ("c_i2", c_i2), ("count", count)):
            _lib.require_gpu_tensor(t, name, torch.int32)
        frame = _Frame(c_n, c_p1, c_p2, c_i1, c_i2)
        out = solve_dynamics(B, nb, maxc, e, count, Mdiag, v, f, rest, fric, frame, Je if e else None, float(dt),
                             eps=opts.get("eps", 1e-12), not_improved_lim=opts.get("not_improved_lim", 3),
                             max_iter=opts.get("max_iter", 10), compute=opts.get("compute", "f64"), pinned=bool(opts.get("pinned", False)))
        ctx.save_for_backward(Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2)
        # what the backward reads - the workspace and the word - WITHOUT the tensor this function returns: autograd stamps its node on the
        # returned tensor, and a ctx that holds that tensor is a cycle through C++ the garbage collector cannot see (node -> ctx -> v_new
        # -> grad_fn -> node): until round 4 every differentiable step leaked its 58 KB-per-scene workspace (profiles/r04_rollout_leak.txt)
        ctx.Je, ctx.dims, ctx.dt, ctx.compute = (Je if e else None), (B, nb, maxc, e), float(dt), opts.get("compute", "f64")
        ctx.out = {k: v for k, v in out.items() if k != "v_new"}
        opts["last"] = out
        return out["v_new"]

    @staticmethod
    def backward(ctx, dl_dv):
        Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2 = ctx.saved_tensors
        B, nb, maxc, e = ctx.dims
        want_Je = bool(e) and ctx.needs_input_grad[11]
        g = solve_dynamics_backward(B, nb, maxc, e, Mdiag, v, f, rest, fric, _Frame(c_n, c_p1, c_p2, c_i1, c_i2), ctx.Je,
                                    ctx.dt, dl_dv, ctx.out, compute=ctx.compute, want_Je=want_Je)
        keys = ("Mdiag", "v", "f", "rest", "fric", "c_n", "c_p1", "c_p2")
        return (tuple(g[k] if need else None for k, need in zip(keys, ctx.needs_input_grad[:8])) + (None, None, None) +
                (g["Je"] if want_Je else None,) + (None, None))


def post_stabilization(B, nb, maxc, e, count, Mdiag, v, rest, cb, Je, p=None, dt_scene=None, dt=0.0, p_out=None,
                       eps=1e-12, not_improved_lim=3, max_iter=10, compute="f64", ws=None, out=None):
    """`PdipmEngine.post_stabilization` (`engines.py:80-116`) for B scenes with per-scene contact counts: one launch
    of `lcp_post_stabilization_f32`.  With `p` / `p_out` (float64 poses) it also makes the correction move of
    `World.step_dt` (`world.py:110-117`): p_out = p + (dp / 2) dt_scene.  Returns dict(dp, iters, status, ws)."""
    lib = _lib.load()
    dev = v.device
    comp = _COMPUTE[compute]
    need = _lib.workspace_bytes(B, 3 * nb, 4 * maxc, e, comp)
    comp |= _lib.path_bits()
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
    if out is None:
        out = {"dp": torch.empty(B, nb, 3, dtype=torch.float32, device=dev),
               "iters": torch.empty(B, dtype=torch.int32, device=dev),
               "status": torch.empty(B, dtype=torch.int32, device=dev)}
    out["ws"] = ws
    for name, t in (("p", p), ("p_out", p_out), ("dt_scene", dt_scene)):
        if t is not None:
            _lib.require_gpu_tensor(t, name, torch.float64)
    P = _lib.ptr
    with _on_device(dev):
        rc = lib.lcp_post_stabilization_f32(B, nb, maxc, e, P(count), P(Mdiag), P(v), P(rest), P(cb.c_n), P(cb.c_p1),
                                            P(cb.c_p2), P(cb.c_i1), P(cb.c_i2), P(Je) if e else None, float(eps),
                                            int(max_iter), int(not_improved_lim), comp, P(p), P(dt_scene), float(dt),
                                            P(p_out), P(out["dp"]), P(out["iters"]), P(out["status"]), P(ws),
                                            _lib.stream_ptr(dev))
    _lib.check(rc, "lcp_post_stabilization_f32")
    out["compute"] = comp
    return out


def post_stabilization_backward(B, nb, maxc, e, Mdiag, v, rest, cb, Je, dl_ddp, out, compute="f64", want_Je=False):
    """Backward of `post_stabilization` with respect to its physical inputs (the reference's autograd through
    `engines.py:80-116` and `lcp.py:37-64`): one launch of `lcp_post_stabilization_backward_f32` on the workspace the forward
    left in `out`.  Returns dict(Mdiag, v [B,nb,3], rest [B,nb], c_n, c_p1, c_p2 [B,maxc,2]) and, with `want_Je`, Je [B,e,3nb]."""
    lib = _lib.load()
    dev = v.device
    dl_ddp = _lib.require_gpu_tensor(dl_ddp.to(torch.float32).contiguous(), "dl_ddp", torch.float32)
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    grads = {"Mdiag": new(B, nb, 3), "v": new(B, nb, 3), "rest": new(B, nb), "c_n": new(B, maxc, 2), "c_p1": new(B, maxc, 2),
             "c_p2": new(B, maxc, 2)}
    if want_Je and e:
        grads["Je"] = new(B, e, 3 * nb)
    P = _lib.ptr
    with _on_device(dev):
        rc = lib.lcp_post_stabilization_backward_f32(B, nb, maxc, e, P(Mdiag), P(v), P(rest), P(cb.c_n), P(cb.c_p1), P(cb.c_p2),
                                                     P(cb.c_i1), P(cb.c_i2), P(Je) if e else None, P(dl_ddp), _word(out, compute),
                                                     P(grads["Mdiag"]), P(grads["v"]), P(grads["rest"]), P(grads["c_n"]),
                                                     P(grads["c_p1"]), P(grads["c_p2"]), P(grads["Je"]) if (want_Je and e) else None,
                                                     P(out["ws"]), _lib.stream_ptr(dev))
    _lib.check(rc, "lcp_post_stabilization_backward_f32")
    return grads


class PostStabilizationFunction(torch.autograd.Function):
    """dp = PostStabilizationFunction.apply(Mdiag, v, rest, c_n, c_p1, c_p2, c_i1, c_i2, count, Je, opts) - the engine's
    `post_stabilization` (`engines.py:80-116`) as an autograd node: forward `lcp_post_stabilization_f32` (no pose update),
    backward `lcp_post_stabilization_backward_f32`.  All tensors float32 on the GPU; every call owns its workspace."""

    @classmethod
    def apply(cls, Mdiag, v, rest, c_n, c_p1, c_p2, c_i1, c_i2, count, Je, opts):
        # (as SolveDynamicsFunction.apply: a recorded correction of a size without a fused backward goes through the dense boundary)
        if count is None:
            count = torch.full((v.shape[0],), c_n.shape[1], dtype=torch.int32, device=v.device)
        args = (Mdiag, v, rest, c_n, c_p1, c_p2, c_i1, c_i2, count, Je, opts)
        if torch.is_grad_enabled() and any(isinstance(a, torch.Tensor) and a.requires_grad for a in args):
            nb, maxc = v.shape[1], c_n.shape[1]
            e = 0 if Je is None or Je.numel() == 0 else Je.shape[1]
            if not _lib.load().lcp_post_stabilization_has_backward(nb, maxc, e, _COMPUTE[opts.get("compute", "f64")] | _lib.path_bits()):
                from .dense_step import post_stabilization_dense
                return post_stabilization_dense(*args)
        return super(PostStabilizationFunction, cls).apply(*args)

    @staticmethod
    def forward(ctx, Mdiag, v, rest, c_n, c_p1, c_p2, c_i1, c_i2, count, Je, opts):
        B, nb = v.shape[0], v.shape[1]
        maxc = c_n.shape[1]
        e = 0 if Je is None or Je.numel() == 0 else Je.shape[1]
        for name, t in (("Mdiag", Mdiag), ("v", v), ("rest", rest), ("c_n", c_n), ("c_p1", c_p1), ("c_p2", c_p2)):
            _lib.require_gpu_tensor(t, name, torch.float32)
        frame = _Frame(c_n, c_p1, c_p2, c_i1, c_i2)
        # opts["ps_pose"] = (p, dt_scene, p_out): the kernel also makes the correction move of world.py:110-117, p_out = p + (dp / 2) dt_scene
        pose = opts.get("ps_pose") or (None, None, None)
        out = post_stabilization(B, nb, maxc, e, count, Mdiag, v, rest, frame, Je if e else None, p=pose[0], dt_scene=pose[1], p_out=pose[2],
                                 compute=opts.get("compute", "f64"))
        ctx.save_for_backward(Mdiag, v, rest, c_n, c_p1, c_p2, c_i1, c_i2)
        ctx.Je, ctx.dims, ctx.compute = (Je if e else None), (B, nb, maxc, e), opts.get("compute", "f64")
        ctx.out = {k: v for k, v in out.items() if k != "dp"}          # (not the returned tensor: see SolveDynamicsFunction)
        opts["last_post_stab"] = out
        return out["dp"]

    @staticmethod
    def backward(ctx, dl_ddp):
        Mdiag, v, rest, c_n, c_p1, c_p2, c_i1, c_i2 = ctx.saved_tensors
        B, nb, maxc, e = ctx.dims
        want_Je = bool(e) and ctx.needs_input_grad[9]
        g = post_stabilization_backward(B, nb, maxc, e, Mdiag, v, rest, _Frame(c_n, c_p1, c_p2, c_i1, c_i2), ctx.Je, dl_ddp,
                                        ctx.out, compute=ctx.compute, want_Je=want_Je)
        keys = ("Mdiag", "v", "rest", "c_n", "c_p1", "c_p2")
        return (tuple(g[k] if need else None for k, need in zip(keys, ctx.needs_input_grad[:6])) + (None, None, None) +
                (g["Je"] if want_Je else None,) + (None,))


class ContactWorld:
    """B independent scenes WITH contact detection, advanced together on one GPU: the batched counterpart of
    the reference's `World` (`physics/world.py:17-122`) with its default `DiffContactHandler` and `PdipmEngine`.

        world = ContactWorld(geom, p, v, Mdiag, f, rest, fric, Je=None, dt=1/30)
        world.step()        # two launches, no host synchronisation:
                            #   lcp_solve_dynamics_f32      new_v            (engines.py:26-78)
                            #   lcp_move_find_contacts_f64  p, contacts, t   (world.py:88-101,122,139-142; contacts.py)

    State: `p` [B,nb,3] float64 (rot, x, y), `v` [B,nb,3] float32, `t` [B] float64 (scenes advance by their own
    accepted dt, world.py:122), `contacts` (`contacts.ContactBuffers`: padded records + `count`).
    Forces: a constant `f` or `force_fn(t)` (time-dependent, `forces.py:29-48`).  `step(differentiable=True)` records the
    step in torch's autograd graph (roll-out gradients, `demos/grad_demo.py`).  Joints: a constant Jacobian `Je`
    (Total/X/Y/Rot constraints) or a `JointSet` (revolute / fixed joints, Jacobian rebuilt every step and differentiated);
    at most 32 bodies per scene.  Scenes with 3 nb <= 32, maxc <= 16, e <= 4 run on the four-scenes-per-wave solver, up to
    64 contacts and 24 equality rows (3 nb + e <= 56) on the wave-per-scene body-space solver (both with a fused backward),
    anything else on the generic kernels; a differentiable step of such a size goes through the dense boundary
    (physics/dense_step.py: torch assembly on the device + `LCPFunction`, the reference's own route).
    `post_stab=True` (off by default, as in the reference: utils.py:30) adds the two launches of world.py:109-121 to a
    step: `lcp_post_stabilization_f32` (frictionless LCP + correction move) and a contact re-detection.
    """

    def __init__(self, geom, p, v, Mdiag, f, rest, fric, Je=None, dt=1.0 / 30, eps=0.1, tol=1e-6,
                 strict_no_penetration=True, maxc=16, max_iter=10, compute="f64", solver_eps=1e-12,
                 not_improved_lim=3, max_trials=64, check=True, post_stab=False, force_fn=None, joints=None):
        from . import contacts as _contacts
        self.post_stab = bool(post_stab)
        # `force_fn(t) -> f [B,nb,3] float32` replaces the constant force: the batched form of the reference's
        # `ExternalForce(force_func)` (forces.py:29-48, `force_func(t)` evaluated at the world's clock every step,
        # world.py:135-137); `t` is the per-scene clock [B] float64 on the device - build f with tensor operations
        # (e.g. `torch.where(t < 0.1, ...)`), no host synchronisation.
        self.force_fn = force_fn
        self._ps_out = self._ps_ws = None
        self._Je_spare = None
        self._phase, self._graphs = 0, {}
        self._contacts_mod = _contacts
        self.geom = geom
        dev = p.device
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        self.p = p.to(dtype=torch.float64).contiguous()
        self.v, self.Mdiag, self.f, self.rest, self.fric = f32(v), f32(Mdiag), f32(f), f32(rest), f32(fric)
        self.B, self.nb = self.p.shape[0], self.p.shape[1]
        # `joints` (a `joints.JointSet`): joints whose Jacobian follows the pose - the reference's revolute `Joint` and
        # `FixedJoint` (constraints.py:13-92) next to the X / Y / Rot / Total constraints; Je is then rebuilt on the device
        # after every move (`lcp_joint_jacobian_f64`).  `Je`: a constant Jacobian instead.
        self.joints = joints
        if joints is not None:
            if Je is not None:
                raise ValueError("give either a constant Je or a JointSet")
            self.e = joints.e
            self.Je = joints.jacobian(self.p)
            self._jrot0 = joints.jrot1.clone()                              # (for `restart`)
        else:
            self.e = 0 if Je is None or Je.numel() == 0 else Je.shape[1]
            self.Je = f32(Je) if self.e else None
        # a constant Je that pins the leading coordinates (the fixed floor of the demo worlds) or no Je at all: LCP_HINT_PINNED,
        # checked once here on the host
        self._pinned = joints is None and rows_pin_leading_coordinates(self.Je)
        self.dt, self.eps, self.tol, self.strict = float(dt), float(eps), float(tol), bool(strict_no_penetration)
        self.maxc, self.max_iter, self.compute = int(maxc), int(max_iter), compute
        self.solver_eps, self.lim, self.max_trials = solver_eps, not_improved_lim, max_trials
        self.t = torch.zeros(self.B, dtype=torch.float64, device=dev)
        self._xy_mask = (torch.arange(3, device=dev) > 0).reshape(1, 1, 3)    # (rot, x, y): the translation columns (device op: capturable)
        self._ws = self._out = None
        self.check = bool(check)
        # OR of every step's per-scene status bits (device side, no synchronisation): LCP_ST_TRUNCATED in here means a
        # scene once had more contacts than `maxc` and was solved with a cut list - `assert_not_truncated()` / `run()`
        self.sticky_status = torch.zeros(self.B, dtype=torch.int32, device=dev)
        # world.py:65-66: contacts of the initial pose; :67-70: refuse interpenetration at start
        self.contacts = _contacts.find_contacts(geom, self.p, maxc=self.maxc, eps=self.eps)
        if check:
            self.check_capacity()
            if self.strict and bool((self.contacts.max_pen > self.tol).any()):
                raise AssertionError("Interpenetration at start (world.py:68-70)")

    def restart(self, p, v=None, jrot1=None, t=0.0):
        """Put the world back at pose `p` [B,nb,3] with velocities `v` (default: at rest), joint angles `jrot1` (default: those
        the `JointSet` had when the world was built) and clock `t`: contacts re-detected, joint Jacobian rebuilt - device work
        only (no host check), so that a roll-out which starts with `restart` can be captured into a HIP graph.  `p` / `v` may
        require grad (differentiable steps); `Mdiag`, `f`, `rest`, `fric`, `force_fn` are plain attributes and can be reassigned."""
        own = lambda x, dt_: x.to(dt_).contiguous() if x.requires_grad else x.detach().to(dt_).clone().contiguous()
        self.p = own(p, torch.float64)
        self.v = torch.zeros(self.B, self.nb, 3, dtype=torch.float32, device=self.p.device) if v is None else own(v, torch.float32)
        self.t.fill_(float(t))
        self.sticky_status.zero_()              # (a device op: an overflow of an earlier roll-out must not fail this one's check)
        self._p_geom_src = self._jrot_src = None
        self._graphs, self._phase = {}, 0
        pd = self.p.detach()
        if self.joints is not None:
            self.joints.jrot1.copy_(self._jrot0 if jrot1 is None else jrot1)
            self.Je = self.joints.jacobian(pd)
        self.contacts = self._contacts_mod.find_contacts(self.geom, pd, maxc=self.maxc, eps=self.eps)
        return self

    def check_capacity(self):
        """Host check (synchronises): no scene produced more than `maxc` contacts."""
        worst = int(self.contacts.count.max())
        if worst > self.maxc:
            raise RuntimeError("a scene has %d contacts but maxc = %d" % (worst, self.maxc))

    def truncated_scenes(self):
        """Scenes that at some step had more than `maxc` contacts (synchronises)."""
        return torch.nonzero(self.sticky_status & _lib.ST_TRUNCATED).flatten()

    def assert_not_truncated(self):
        """Host check (synchronises): the reference's contact list is unbounded (world.py:139-142), ours is capped at
        `maxc`; a scene that overflowed was solved without its surplus contacts and the run is not the reference's."""
        bad = self.truncated_scenes()
        if bad.numel():
            raise RuntimeError("%d scene(s) exceeded maxc = %d contacts during the run (first: scene %d): their "
                               "contact lists were truncated - rebuild the world with a larger maxc"
                               % (bad.numel(), self.maxc, int(bad[0])))

    def run(self, nsteps, graph=True):
        """`nsteps` calls of `step()`.  With `graph=True` the launches of TWO consecutive steps (after two steps the
        double-buffered state tensors are back in their slots) are captured once into a HIP graph and replayed: the
        host then issues one graph launch per two simulation steps instead of 2-4 kernel launches with their Python
        marshalling per step - what bounds small batches.  Same kernels, same order, same results."""
        k = 0
        if graph and nsteps >= 2:
            while not self._graphs and k < 2:
                self.step()                                                # warm-up: every buffer exists before capture
                k += 1
            g = self._graphs.get(self._phase)
            if g is None and nsteps - k >= 2:
                torch.cuda.synchronize(self.p.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.step()
                    self.step()
                self._graphs[self._phase] = g                              # (capture only records: nothing has run)
            while g is not None and nsteps - k >= 2:
                g.replay()
                k += 2
        while k < nsteps:
            self.step()
            k += 1
        if self.check:
            self.assert_not_truncated()                                    # one synchronisation per run, none per step

    def step_autograd(self):
        """`step()` as a node of torch's autograd graph, for losses evaluated after a roll-out (`demos/grad_demo.py:45-50`,
        `experiments/inference.py:55-61`).  Same two launches; the graph of one step is

            frame   = ContactFrameFunction(p)                         backward: lcp_contact_frame_backward_f64
            v_new   = SolveDynamicsFunction(Mdiag, v, f(t), rest, fric, frame, ...)   backward: lcp_step_backward_f32
            p_new   = p + v_new dt_used                               (bodies.py:80-82; dt_used: the dt the step_dt loop accepted)

        `Mdiag, f, rest, fric, v, p` (and what `force_fn` closes over) may require grad.  State tensors are replaced, not
        overwritten, and every step keeps its own workspace and contact snapshot for the backward.  The joint Jacobian of
        revolute / fixed joints is differentiated through the pose and the joint angle (dL/dJe: lcp_step_backward_je_f32);
        with `post_stab` the correction move is one more node (`PostStabilizationFunction`, world.py:109-121)."""
        ct = self._contacts_mod
        # the contact buffers at the current pose become this step's frame (kept for its backward); the detection below fills a
        # fresh set - no copies
        frame = self.contacts
        frame.retired = True                                               # (nothing writes these records again: no copy in ContactFrameFunction)
        cb = self.contacts = ct.ContactBuffers(self.B, self.nb, self.maxc, self.p.device)
        self._autograd_owned = True
        # HIP graphs captured by an earlier run(graph=True) hold raw pointers to the buffers retired here (they now belong to this
        # step's backward) and to the old state tensors: a later run() must capture again on the new ones
        self._graphs, self._phase = {}, 0
        # the pose the GEOMETRY is differentiated at: the same values as self.p, but a rotation increment that is exactly
        # zero carries no gradient - the reference turns its hulls' vertices by the increment and skips the turn when the
        # increment is zero (bodies.py:199-202 `if rot.item() != 0: self.rotate_verts(rot)`), so its autograd has no
        # vertex path through such a step (a hull in free fall with no torque on it, for instance)
        p_geo = self._p_geom if getattr(self, "_p_geom_src", None) is self.p else self.p
        c_n, c_p1, c_p2 = ct.ContactFrameFunction.apply(p_geo, self.geom, frame, self.eps)
        f = self.f if self.force_fn is None else self.force_fn(self.t).to(torch.float32).contiguous()
        opts = {"max_iter": self.max_iter, "eps": self.solver_eps, "not_improved_lim": self.lim, "compute": self.compute,
                "pinned": self._pinned}
        Je = self.Je
        js = self.joints
        if js is not None and js.pose_dependent:
            # the joint Jacobian as a function of the pose and of the revolute joints' angles (constraints.py:26-50): the
            # kernel's values, the gradient of the torch expression; lcp_step_backward_je_f32 returns dL/dJe
            if getattr(self, "_jrot_src", None) is not self.p:
                self._jrot_ad, self._jrot_src = js.jrot1.clone(), self.p
            Je = _JointJacobianFn.apply(self.p, self._jrot_ad, js.jr1, js, self.Je)
        v_new = SolveDynamicsFunction.apply(self.Mdiag, self.v.contiguous(), f, self.rest, self.fric, c_n, c_p1, c_p2, frame.c_i1,
                                            frame.c_i2, frame.count, Je, self.dt, opts)
        out = opts["last"]
        torch.bitwise_or(self.sticky_status, out["status"], out=self.sticky_status)
        p_start = self.p
        ct.move_and_find_contacts(self.geom, p_start.detach(), v_new.detach(), self.dt, eps=self.eps, tol=self.tol,
                                  strict=self.strict, dt_floor=self.dt / 4, max_trials=self.max_trials, t=self.t, out=cb)
        # the accepted pose: the kernel's value, the gradient of p + v dt_used (a zero rotation increment carries no gradient to the geometry)
        self.v = v_new
        if js is not None:                                                 # joint.move(dt): rot1 += body1.v[0] dt (constraints.py:39-43)
            self.Je = js.jacobian(cb.p_out, v=v_new.detach().contiguous(), dt_scene=cb.dt_used)
        if js is not None and js.pose_dependent:
            self.p, self._p_geom, self._jrot_ad = _StateUpdate.apply(p_start, p_geo, self._jrot_ad, v_new, cb.p_out, js.jrot1.clone(),
                                                                     cb.dt_used, 1.0, js)
            self._jrot_src = self.p
        else:
            self.p, self._p_geom = _StateUpdate.apply(p_start, p_geo, None, v_new, cb.p_out, None, cb.dt_used, 1.0, None)
        self._p_geom_src = self.p
        if self.post_stab:
            # world.py:109-121: dp = engine.post_stabilization(world) at the moved pose with the contacts found there and the
            # NEW velocities; dp /= 2; the bodies (and the joints) move by dp dt; contacts are detected again
            frame2 = cb                                                    # (retired like `frame`: the detection at the end fills a fresh set)
            frame2.retired = True
            dt_used = cb.dt_used
            g_n, g_p1, g_p2 = ct.ContactFrameFunction.apply(self._p_geom, self.geom, frame2, self.eps)
            pose_dep = js is not None and js.pose_dependent
            Je2 = _JointJacobianFn.apply(self.p, self._jrot_ad, js.jr1, js, self.Je) if pose_dep else self.Je
            p_corr = torch.empty_like(cb.p_out)                            # the corrected pose: the kernel's own move, as in step()
            opts["ps_pose"] = (cb.p_out, dt_used, p_corr)
            dp_s = PostStabilizationFunction.apply(self.Mdiag, v_new.contiguous(), self.rest, g_n, g_p1, g_p2, frame2.c_i1, frame2.c_i2,
                                                   frame2.count, Je2, opts)
            ps = opts["last_post_stab"]
            torch.bitwise_or(self.sticky_status, ps["status"], out=self.sticky_status)
            p_mid, g_mid = self.p, self._p_geom
            if js is not None:                                             # the joints follow the correction move (world.py:112-116)
                self.Je = js.jacobian(p_corr, v=dp_s.detach().contiguous(), dt_scene=dt_used, vscale=0.5)
            if pose_dep:
                self.p, self._p_geom, self._jrot_ad = _StateUpdate.apply(p_mid, g_mid, self._jrot_ad, dp_s, p_corr, js.jrot1.clone(), dt_used,
                                                                         0.5, js)
                self._jrot_src = self.p
            else:
                self.p, self._p_geom = _StateUpdate.apply(p_mid, g_mid, None, dp_s, p_corr, None, dt_used, 0.5, None)
            self._p_geom_src = self.p
            self.contacts = ct.find_contacts(self.geom, p_corr, maxc=self.maxc, eps=self.eps)   # world.py:121
            out = dict(out)
            out["post_stab"] = ps
        ret = dict(out)
        ret["v_new"] = v_new
        return ret

    def step(self, differentiable=False):
        """`World.step()` = `step_dt(self.dt)` (`world.py:72-122`) for every scene."""
        if differentiable:
            return self.step_autograd()
        if getattr(self, "_autograd_owned", False):
            # the state tensors are outputs of the autograd graph (their storage belongs to the retired contact buffers of the last
            # differentiable step): the double buffering below must not hand them to a kernel as an output slot
            # - and the contact buffers of the last differentiable step (dt_used, the accepted pose) are saved in that graph
            self.p, self.v, self._autograd_owned = self.p.detach().clone(), self.v.detach().clone(), False
            self.contacts = self.contacts.clone()
            self._graphs, self._phase = {}, 0                            # (no graph captured before these buffers existed may replay)
        self._phase ^= 1
        cb = self.contacts
        f = self.f if self.force_fn is None else self.force_fn(self.t).to(torch.float32).contiguous()
        out = solve_dynamics(self.B, self.nb, self.maxc, self.e, cb.count, self.Mdiag, self.v, f, self.rest,
                             self.fric, cb, self.Je, self.dt, eps=self.solver_eps, not_improved_lim=self.lim,
                             max_iter=self.max_iter, compute=self.compute, ws=self._ws, out=self._out, pinned=self._pinned)
        self._ws, self._out = out["ws"], out
        torch.bitwise_or(self.sticky_status, out["status"], out=self.sticky_status)
        self.v, out["v_new"] = out["v_new"], self.v                      # world.py:87 set_v(new_v)
        self._contacts_mod.move_and_find_contacts(self.geom, self.p, self.v, self.dt, eps=self.eps, tol=self.tol,
                                                  strict=self.strict, dt_floor=self.dt / 4,
                                                  max_trials=self.max_trials, t=self.t, out=cb)
        self.p, cb.p_out = cb.p_out, self.p                              # accepted pose becomes the state (double buffer)
        if self.joints is not None:                                      # joint.move(dt) + the Jacobian at the new pose (world.py:91-92)
            self._Je_spare = self.joints.jacobian(self.p, v=self.v, dt_scene=cb.dt_used, out=self._Je_spare)
            self.Je, self._Je_spare = self._Je_spare, self.Je
        if self.post_stab:                                               # world.py:109-121
            # the engine's defaults here (engines.py:114 `self.lcp_solver()`), not the dynamics solve's settings;
            # the pose is corrected in place (every thread reads and writes its own entry)
            ps = post_stabilization(self.B, self.nb, self.maxc, self.e, cb.count, self.Mdiag, self.v, self.rest, cb,
                                    self.Je, p=self.p, dt_scene=cb.dt_used, dt=self.dt, p_out=self.p,
                                    compute=self.compute, ws=self._ps_ws, out=self._ps_out)
            self._ps_ws, self._ps_out = ps["ws"], ps
            if self.joints is not None:                                  # the joints follow the correction move too (world.py:112-116)
                self._Je_spare = self.joints.jacobian(self.p, v=ps["dp"], dt_scene=cb.dt_used, vscale=0.5, out=self._Je_spare)
                self.Je, self._Je_spare = self._Je_spare, self.Je
            self._contacts_mod.find_contacts(self.geom, self.p, maxc=self.maxc, eps=self.eps, out=cb)   # world.py:121
            out["post_stab"] = ps
        ret = dict(out)
        ret["v_new"], ret["v_prev"] = self.v, out["v_new"]               # the NEW velocities; the spare buffer holds the old
        return ret

    def get_v(self):
        return self.v

    def get_p(self):
        return self.p

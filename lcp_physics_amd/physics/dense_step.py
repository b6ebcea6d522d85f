"""The contact-list entry points THROUGH THE DENSE BOUNDARY, for the sizes whose fused kernels keep nothing a backward could read.

`SolveDynamicsFunction` / `PostStabilizationFunction` (batched_world.py) differentiate a step with one fused backward launch where
there is one: since round 6 every size the forward kernels accept (`lcp_step_bwd_kernel` on the generic kernels beyond the one-wave
sizes), except the wave64 step family (fp32 arithmetic, 3 nb <= 16, 5..8 joint rows).  What is left - and this module as the A/B
partner of the fused route in the tests - takes the route the reference itself takes (`engines.py:26-116`): the LCP is ASSEMBLED as dense tensors by
differentiable torch operations on the device (`world.py:144-234` -> `engines.py:50-74`), solved by `LCPFunction` (the HIP kernels of
`lcp/lcp.py`, any size; forward `pdipm.py:24-199`, backward `lcp.py:37-64`) and torch's autograd carries the dense gradients back to
the physical inputs.  Slower than a fused step by the dense traffic (B x 4 nc x 3 nb matrices are written, read and differentiated),
equal in result: in fp64 arithmetic (`compute="f64"`) the tensors are assembled and solved in float64, the reference's own dtype.

Per-scene contact counts: the reference solves each scene with exactly its own contacts, so scenes are grouped by count (one host
read of `count`) and every group is solved at its own size; scenes without contacts take the linear solve of `engines.py:36-49`.

No CPU path: every tensor stays on the GPU and the solve is `liblcp_hip.so`'s (it raises without the library)."""
import torch

from .. import _lib
from ..lcp.lcp import LCPFunction


def _cross(a, b):
    return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]                   # physics/utils.py:93-96


def contact_jacobians(c_n, c_p1, c_p2, c_i1, c_i2, nb):
    """`World.Jc()` / `World.Jf()` (`world.py:172-211`) for B scenes: Jc [B,nc,3nb], Jf [B,2nc,3nb] (rows 2c, 2c+1: the two
    friction directions of contact c), differentiable in n, p1, p2."""
    B, nc, _ = c_n.shape
    t = torch.stack([c_n[..., 1], -c_n[..., 0]], dim=-1)                  # left_orthogonal (utils.py:99-102)
    row = lambda r, d: torch.stack([_cross(r, d), d[..., 0], d[..., 1]], dim=-1)
    at = lambda i: i.long().reshape(B, nc, 1, 1).expand(B, nc, 1, 3)
    zero = c_n.new_zeros(B, nc, nb, 3)
    # body 2 is written after body 1 (plain assignments in the reference)
    Jc = zero.scatter(2, at(c_i1), row(c_p1, c_n).unsqueeze(2)).scatter(2, at(c_i2), -row(c_p2, c_n).unsqueeze(2))
    Jt = zero.scatter(2, at(c_i1), row(c_p1, t).unsqueeze(2)).scatter(2, at(c_i2), -row(c_p2, t).unsqueeze(2))
    Jf = torch.stack([Jt, -Jt], dim=2)                                     # dir2 = -dir1 (world.py:196-210)
    return Jc.reshape(B, nc, 3 * nb), Jf.reshape(B, 2 * nc, 3 * nb)


def _pair_mean(x, c_i1, c_i2):
    return 0.5 * (torch.gather(x, 1, c_i1.long()) + torch.gather(x, 1, c_i2.long()))   # world.py:144-151, :213-224


def assemble_dynamics(Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2, Je, dt):
    """(Q, p, G, h, A, b, F) of `engines.py:31-32,50-74` (contact branch), dense and differentiable."""
    B, nb, _ = v.shape
    nc, nz = c_n.shape[1], 3 * nb
    Md, vv = Mdiag.reshape(B, nz), v.reshape(B, nz)
    u = Md * vv + dt * f.reshape(B, nz)                                    # engines.py:32
    Jc, Jf = contact_jacobians(c_n, c_p1, c_p2, c_i1, c_i2, nb)
    h = torch.cat([(Jc @ vv.unsqueeze(2)).squeeze(2) * _pair_mean(rest, c_i1, c_i2), v.new_zeros(B, 3 * nc)], dim=1)   # :53, :74
    G = torch.cat([Jc, Jf, v.new_zeros(B, nc, nz)], dim=1)                 # :67-68
    ar = torch.arange(nc, device=v.device)
    F = v.new_zeros(B, 4 * nc, 4 * nc)
    F[:, nc + 2 * ar, 3 * nc + ar] = 1                                     # :70  E
    F[:, nc + 2 * ar + 1, 3 * nc + ar] = 1
    F[:, 3 * nc + ar, nc + 2 * ar] = -1                                    # :72-73  -E^T
    F[:, 3 * nc + ar, nc + 2 * ar + 1] = -1
    F[:, 3 * nc + ar, ar] = _pair_mean(fric, c_i1, c_i2)                   # :71  mu
    A, b = (Je, v.new_zeros(B, Je.shape[1])) if Je is not None else (v.new_zeros(0), v.new_zeros(0))
    return torch.diag_embed(Md), u, G, h, A, b, F


def assemble_post_stabilization(Mdiag, v, rest, c_n, c_p1, c_p2, c_i1, c_i2, Je):
    """The frictionless LCP of `engines.py:80-116`: Q = M, p = 0, G = Jc, h = Jc v - (Jc v) restitution, A = Je, b = Je v, F = 0."""
    B, nb, _ = v.shape
    nc, nz = c_n.shape[1], 3 * nb
    vv = v.reshape(B, nz)
    Jc, _ = contact_jacobians(c_n, c_p1, c_p2, c_i1, c_i2, nb)
    jv = (Jc @ vv.unsqueeze(2)).squeeze(2)
    gc = jv + jv * -_pair_mean(rest, c_i1, c_i2)                           # :87-89
    A, b = (Je, (Je @ vv.unsqueeze(2)).squeeze(2)) if Je is not None else (v.new_zeros(0), v.new_zeros(0))
    return torch.diag_embed(Mdiag.reshape(B, nz)), v.new_zeros(B, nz), Jc, gc, A, b, v.new_zeros(B, nc, nc)


def _linear(Md, top, Je, bottom):
    """No contacts (`engines.py:36-49`, `:91-103`): x of [[M, -Je^T], [Je, 0]] x = [top; bottom], by the Schur complement on Je."""
    if Je is None:
        return top / Md
    JM = Je / Md.unsqueeze(1)                                              # Je M^-1
    S = JM @ Je.transpose(1, 2)
    y = torch.linalg.solve(S, (bottom - (JM @ top.unsqueeze(2)).squeeze(2)).unsqueeze(2))
    return (top + (Je.transpose(1, 2) @ y).squeeze(2)) / Md


def _groups(count, maxc, B, dev):
    """[(contacts, scene indices or None)]: all scenes at `maxc` without counts, otherwise one group per distinct count."""
    if count is None:
        return [(maxc, None)], None
    cnt = count.clamp(0, maxc)
    host = cnt.cpu()
    if bool((host == maxc).all()):
        return [(maxc, None)], count > maxc
    return [(int(c), (cnt == int(c)).nonzero().flatten()) for c in host.unique().tolist()], count > maxc


_WARNED = set()


def _warn_once(kind, nb, maxc, e):
    """The switch to this route used to be silent: said once per (entry, size) and process."""
    key = (kind, nb, maxc, e)
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn("lcp_physics_amd: a recorded %s step of %d bodies / %d contacts / %d joint rows has no fused backward kernel "
                      "(lcp_step_has_backward / lcp_post_stabilization_has_backward): it goes through the dense LCPFunction boundary (physics/dense_step.py - the reference's "
                      "own route: correct and differentiable, not fast; one host synchronisation per step for the contact counts)"
                      % (kind, nb, maxc, e), RuntimeWarning, stacklevel=4)


def _run(kind, phys, lists, count, Je, dt, opts):
    """Shared driver: phys = per-body tensors, lists = (c_n, c_p1, c_p2, c_i1, c_i2).  Returns (x [B,nb,3] float32, record)."""
    c_n = lists[0]
    v = phys["v"]
    B, nb, _ = v.shape
    maxc, dev = c_n.shape[1], v.device
    e = 0 if Je is None or Je.numel() == 0 else Je.shape[1]
    f64 = opts.get("compute", "f64") == "f64"
    dd = torch.float64 if f64 else torch.float32
    up = lambda t: None if t is None else t.to(dd)
    nrows = 4 * maxc if kind == "dynamics" else maxc
    rec = {"z": torch.zeros(B, nrows, dtype=torch.float32, device=dev), "s": torch.zeros(B, nrows, dtype=torch.float32, device=dev),
           "y": torch.zeros(B, e, dtype=torch.float32, device=dev) if e else None,
           "iters": torch.zeros(B, dtype=torch.int32, device=dev), "status": torch.zeros(B, dtype=torch.int32, device=dev),
           "ws": None, "compute": None, "dense_boundary": True}      # (no fused workspace: `require_fused_record` below is what the helpers that need one call)
    _warn_once(kind, nb, maxc, e)
    groups, truncated = _groups(count, maxc, B, dev)
    x_all = None
    for nc, sel in groups:
        take = (lambda t: t) if sel is None else (lambda t: None if t is None else t.index_select(0, sel))
        ph = {k: up(take(t)) for k, t in phys.items()}
        Jg = up(take(Je if e else None))
        Md = ph["Mdiag"].reshape(-1, 3 * nb)
        if nc == 0:
            if kind == "dynamics":
                top = Md * ph["v"].reshape(-1, 3 * nb) + dt * ph["f"].reshape(-1, 3 * nb)
                x = _linear(Md, top, Jg, None if Jg is None else top.new_zeros(top.shape[0], e))
            else:
                vv = ph["v"].reshape(-1, 3 * nb)
                x = -_linear(Md, torch.zeros_like(vv), Jg, None if Jg is None else (Jg @ vv.unsqueeze(2)).squeeze(2)) \
                    if Jg is not None else torch.zeros_like(vv)
        else:
            cn, cp1, cp2 = (up(take(t)[:, :nc]) for t in lists[:3])
            ci1, ci2 = (take(t)[:, :nc] for t in lists[3:])
            if kind == "dynamics":
                lcp = assemble_dynamics(ph["Mdiag"], ph["v"], ph["f"], ph["rest"], ph["fric"], cn, cp1, cp2, ci1, ci2, Jg, dt)
            else:
                lcp = assemble_post_stabilization(ph["Mdiag"], ph["v"], ph["rest"], cn, cp1, cp2, ci1, ci2, Jg)
            solver = LCPFunction(eps=opts.get("eps", 1e-12), not_improved_lim=opts.get("not_improved_lim", 3),
                                 max_iter=opts.get("max_iter", 10), compute=opts.get("compute", "f64"), check=False)
            x = -solver(*[t.contiguous() for t in lcp])                    # engines.py:76-77 / :115
            # the multipliers in the row layout of a capacity-sized LCP ([normal | friction pairs | gamma] blocks), like the fused step
            blocks = ((0, 0, nc), (nc, maxc, 2 * nc), (3 * nc, 3 * maxc, nc)) if kind == "dynamics" else ((0, 0, nc),)
            rows = torch.cat([torch.arange(dst, dst + n, device=dev) for _, dst, n in blocks])
            for name, val in (("z", solver.lams), ("s", solver.slacks)):
                if sel is None:
                    rec[name][:, rows] = val.detach().float()
                else:
                    rec[name][sel.unsqueeze(1), rows.unsqueeze(0)] = val.detach().float()
            scenes = slice(None) if sel is None else sel
            if e:
                rec["y"][scenes] = solver.nus.detach().float()
            rec["iters"][scenes] = solver.iters
            rec["status"][scenes] = solver.status
        x = x.reshape(-1, nb, 3).float()
        x_all = x if sel is None else (x.new_zeros(B, nb, 3) if x_all is None else x_all).index_copy(0, sel, x)
    if truncated is not None:
        rec["status"] |= truncated.to(torch.int32) * _lib.ST_TRUNCATED
    return x_all, rec


def require_fused_record(out, what):
    """Helpers that read a fused step's workspace (`solution_of_step`, `fused_step_backward`, `solve_dynamics_backward`, reuse of
    `out["ws"]`) call this first: a record left by the dense-boundary route has none, and says so instead of dereferencing None."""
    if out is not None and out.get("dense_boundary"):
        raise RuntimeError("%s needs the workspace of a fused step; this record was left by the dense-boundary route "
                           "(physics/dense_step.py: sizes beyond the fused kernels) - differentiate through the returned tensor instead" % what)


def solve_dynamics_dense(Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2, count, Je, dt, opts):
    """`PdipmEngine.solve_dynamics` (`engines.py:26-78`) for B scenes as differentiable torch operations around `LCPFunction`.
    Same arguments as `SolveDynamicsFunction.apply`; returns new_v [B,nb,3] float32 and leaves the record of the solve (z, s, y in the
    fused step's row layout, iters, status) in `opts["last"]`."""
    x, rec = _run("dynamics", {"Mdiag": Mdiag, "v": v, "f": f, "rest": rest, "fric": fric}, (c_n, c_p1, c_p2, c_i1, c_i2), count, Je,
                  float(dt), opts)
    rec["v_new"] = x
    opts["last"] = rec
    return x


def post_stabilization_dense(Mdiag, v, rest, c_n, c_p1, c_p2, c_i1, c_i2, count, Je, opts):
    """`PdipmEngine.post_stabilization` (`engines.py:80-116`) the same way: dp [B,nb,3] float32; record in `opts["last_post_stab"]`.
    With `opts["ps_pose"] = (p, dt_scene, p_out)` the correction move of `world.py:110-117` is made too (p_out = p + (dp / 2) dt_scene)."""
    x, rec = _run("post_stab", {"Mdiag": Mdiag, "v": v, "rest": rest}, (c_n, c_p1, c_p2, c_i1, c_i2), count, Je, 0.0, opts)
    rec["dp"] = x
    pose = opts.get("ps_pose")
    if pose is not None and pose[2] is not None:
        p, dt_scene, p_out = pose
        p_out.copy_(p + (x.detach().double() * 0.5) * dt_scene.reshape(-1, 1, 1))
    opts["last_post_stab"] = rec
    return x

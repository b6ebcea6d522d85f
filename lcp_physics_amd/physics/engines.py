"""Engine plug-ins for the reference's `World` (`lcp_physics/physics/engines.py:11-116`).

`World(bodies, joints, engine=HipPdipmEngine)` works unchanged: the reference instantiates the class with no
arguments (`physics/world.py:26`, `physics/utils.py:142-150`) and calls `engine.solve_dynamics(world, dt) -> new_v`
every step (`world.py:86`) and, when `world.post_stab`, `engine.post_stabilization(world) -> dp` (`world.py:109-111`).

Neither engine assembles a dense LCP on the host.  The world's raw state - diagonal of `M`, velocities, forces,
per-body restitution / friction, the contact list `((normal, p1, p2, penetration), i1, i2)` and the joint Jacobian -
is lifted to the GPU as a batch of ONE scene and handed to the contact-list entry points of the C ABI:

* `HipPdipmEngine` - differentiable.  `solve_dynamics` is one `SolveDynamicsFunction` node (forward
  `lcp_solve_dynamics_f32`: assembly + PDIPM solve, both branches of `engines.py:26-78`; backward
  `lcp_step_backward_f32`: the implicit differentiation of `lcp.py:37-64` contracted through the assembly on chip), so
  `loss.backward()` reaches masses, forces, velocities, restitution / friction coefficients, the contact frame and
  the joint Jacobian `world.Je()` (hence the joints' anchors: `constraints.py:26-50` stays the reference's torch code)
  exactly where the reference's autograd does.  `post_stabilization` (`engines.py:80-116`) is a
  `PostStabilizationFunction` node (forward `lcp_post_stabilization_f32`, backward
  `lcp_post_stabilization_backward_f32`): a `World(post_stab=True)` differentiates as in `experiments/inference.py`.
* `HipFusedEngine` - the same launches without recording a graph (inference).

Sizes: what the contact-list kernels take (64 contacts; for the backward 3 nb + e <= 56 with up to 24 joint rows, or the
four-scenes-per-wave sizes); beyond that the call raises - there is no CPU or dense fallback.
"""
import torch

from . import batched_world
from .batched_world import PostStabilizationFunction, SolveDynamicsFunction


class Engine:
    """Base class for stepping engine (`engines.py:11-14`)."""

    def solve_dynamics(self, world, dt):
        raise NotImplementedError


def _gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("lcp_physics_amd needs a GPU (MI355X); no CPU fallback exists")
    return torch.device("cuda", torch.cuda.current_device())


class _Lifted:
    """What `Engine.solve_dynamics` reads from the reference's `World` (`world.py:124-234`), as float32 GPU tensors with a
    leading batch axis of one.  Built with differentiable torch ops only (cat / reshape / `.to` / slices), so gradients that
    arrive at these tensors flow on to the world's own leaves.

    The reference calls its engine once per `World.step()` with a batch of ONE scene, so what a step costs here is host work, not the
    35 us of the kernel: everything the kernel reads travels in TWO transfers - the floats packed into one tensor on the host (one
    `torch.cat`, one copy, views on the device), the contact indices and the count in one int32 tensor - instead of a dozen small
    copies (round 4: 3.5 ms per step on ball + floor against the reference engine's 0.40; profiles/r05_reference_world_plugin.json)."""

    def __init__(self, world, forces=True):
        dev = _gpu()
        bodies = world.bodies
        nb = len(bodies)
        nz = 3 * nb
        self.nb, self.dev = nb, dev
        Je = world.Je()
        self.e = Je.size(0) if (Je.ndimension() > 1 and Je.numel() > 0) else 0
        contacts = world.contacts or []
        self.nc = len(contacts)
        cap = self.cap = max(1, self.nc)
        flat = lambda t: t.reshape(-1)
        # (one flat list of 1-D tensors -> ONE torch.cat: every intermediate stack / reshape is 1-4 us of host time on a step that
        #  costs the reference's own engine 60 us when nothing touches)
        parts = [flat(world.get_v()), flat(torch.diagonal(world.M()))]
        parts += [b.restitution.reshape(1) for b in bodies]
        parts += [b.fric_coeff.reshape(1) for b in bodies]
        sizes = [nz, nz, nb, nb]
        if forces:
            parts.append(flat(world.apply_forces(world.t)))
            sizes.append(nz)
        if self.e:
            parts.append(flat(Je))                                         # (its gradient flows on to the joints' anchors)
            sizes.append(self.e * nz)
        if contacts:
            for k in range(3):                                             # normal, p1, p2 of ((normal, p1, p2, penetration), i1, i2)
                parts += [c[0][k].reshape(2) for c in contacts]
                sizes.append(2 * cap)
        dtype0 = parts[0].dtype
        if any(q.dtype != dtype0 for q in parts):
            parts = [q if q.dtype == dtype0 else q.to(dtype0) for q in parts]
        host = torch.cat(parts)
        devf = host.to(device=dev, dtype=torch.float32)                    # ONE copy (differentiable: the gradient comes back the same way)
        cut = torch.split(devf, sizes)
        self.v, self.Mdiag = cut[0].view(1, nb, 3), cut[1].view(1, nb, 3)
        self.rest, self.fric = cut[2].view(1, nb), cut[3].view(1, nb)
        i = 4
        self.f = None
        if forces:
            self.f = cut[i].view(1, nb, 3)
            i += 1
        self.Je = None
        if self.e:
            self.Je = cut[i].view(1, self.e, nz)
            i += 1
        if contacts:
            self.c_n, self.c_p1, self.c_p2 = cut[i].view(1, cap, 2), cut[i + 1].view(1, cap, 2), cut[i + 2].view(1, cap, 2)
            ints = [int(c[1]) for c in contacts] + [int(c[2]) for c in contacts] + [self.nc]
        else:                                       # engines.py:35-49: the no-contact branch, a list of capacity one, count 0
            z = torch.zeros(3, 1, cap, 2, dtype=torch.float32, device=dev)
            self.c_n, self.c_p1, self.c_p2 = z[0], z[1], z[2]
            ints = [0, 0, 0]
        devi = torch.tensor(ints, dtype=torch.int32).to(dev)               # the second (and last) copy
        self.c_i1, self.c_i2, self.count = devi[:cap].reshape(1, cap), devi[cap:2 * cap].reshape(1, cap), devi[2 * cap:]


def _fresh(r, buf):
    """The reference's engines return NEW tensors every step (engines.py:76-78, 114-116) and its world mutates what it gets in place
    (world.py:112 `dp /= 2`) or keeps it (trajectory logs, `tmp_v`).  A world on the same GPU in float32 makes `.to()` a no-op: what comes
    back would be a view of the buffer the engine re-uses for the next solve - so a result that still shares storage with that
    buffer is copied."""
    return r.clone() if r.untyped_storage().data_ptr() == buf.untyped_storage().data_ptr() else r


class HipPdipmEngine(Engine):
    """Engine that uses the MI355X primal-dual interior point LCP solver (differentiable)."""

    differentiable = True

    def __init__(self, max_iter=10, compute="f64"):
        self.max_iter = max_iter
        self.compute = compute
        self.last = None                              # z, s, y, iters, status of the latest solve (device tensors)
        self._buffers = {}                            # (entry, nb, cap, e) -> outputs + workspace of the steps nothing differentiates

    def _options(self):
        return {"max_iter": self.max_iter, "eps": 1e-12, "not_improved_lim": 3, "compute": self.compute}

    def _records(self, s):
        """Does this step have to be a node of an autograd graph ?  Only if something it reads requires a gradient - a plain
        simulation (`run_world`, utils.py; grad mode is on there, nothing requires grad) launches the kernel directly on buffers the
        engine keeps: no autograd node, no allocation per step."""
        return (self.differentiable and torch.is_grad_enabled() and
                any(t is not None and t.requires_grad for t in (s.Mdiag, s.v, s.f, s.rest, s.fric, s.c_n, s.c_p1, s.c_p2, s.Je)))

    def solve_dynamics(self, world, dt):
        base = world.get_v()
        opts = self._options()
        with torch.set_grad_enabled(self.differentiable and torch.is_grad_enabled()):
            s = _Lifted(world)
            if self._records(s):
                new_v = SolveDynamicsFunction.apply(s.Mdiag, s.v, s.f, s.rest, s.fric, s.c_n, s.c_p1, s.c_p2, s.c_i1, s.c_i2,
                                                    s.count, s.Je, float(dt), opts)
                self.last = opts.get("last")
            else:
                key = ("dyn", s.nb, s.cap, s.e)
                old = self._buffers.get(key)
                with torch.no_grad():
                    out = batched_world.solve_dynamics(1, s.nb, s.cap, s.e, s.count, s.Mdiag, s.v, s.f, s.rest, s.fric, s, s.Je, float(dt),
                                                       eps=opts["eps"], not_improved_lim=opts["not_improved_lim"], max_iter=opts["max_iter"],
                                                       compute=opts["compute"], ws=None if old is None else old["ws"], out=old)
                self._buffers[key] = self.last = out
                new_v = out["v_new"]
            return _fresh(new_v.reshape(-1).to(device=base.device, dtype=base.dtype), new_v)

    def post_stabilization(self, world):
        base = world.get_v()
        opts = self._options()
        with torch.set_grad_enabled(self.differentiable and torch.is_grad_enabled()):
            s = _Lifted(world, forces=False)
            if self._records(s):
                dp = PostStabilizationFunction.apply(s.Mdiag, s.v, s.rest, s.c_n, s.c_p1, s.c_p2, s.c_i1, s.c_i2, s.count, s.Je, opts)
            else:
                key = ("post", s.nb, s.cap, s.e)
                old = self._buffers.get(key)
                with torch.no_grad():
                    # (the solver's defaults, as PostStabilizationFunction and engines.py:114 `self.lcp_solver()` - not the dynamics solve's settings)
                    out = batched_world.post_stabilization(1, s.nb, s.cap, s.e, s.count, s.Mdiag, s.v, s.rest, s, s.Je,
                                                           compute=opts["compute"], ws=None if old is None else old["ws"], out=old)
                self._buffers[key] = out
                dp = out["dp"]
            return _fresh(dp.reshape(-1).to(device=base.device, dtype=base.dtype), dp)


class HipFusedEngine(HipPdipmEngine):
    """The same device path without autograd bookkeeping (inference): nothing of the step is kept for a backward."""

    differentiable = False

"""Engine plug-ins for the reference's `World` (`lcp_physics/physics/engines.py:11-116`).

`World(bodies, joints, engine=HipPdipmEngine)` works unchanged: the reference instantiates the
class with no arguments (`physics/world.py:26`, `physics/utils.py:142-150`) and calls
`engine.solve_dynamics(world, dt) -> new_v` every step (`world.py:86`).

* `HipPdipmEngine` - differentiable.  Builds (M, u, G, h, Je, b, F) with the same torch
  expressions as `engines.py:50-74` (so autograd reaches masses, forces and contact geometry)
  and solves the mixed LCP with the HIP `LCPFunction`.
* `HipFusedEngine` - inference.  Hands the raw contact list to the device entry points of the batched
  path (assembly + solve in one launch, both branches of `solve_dynamics`, and `post_stabilization`);
  not differentiable.

In the differentiable `HipPdipmEngine` the no-contact branch (`engines.py:35-49`, an equality-only linear
solve autograd must see through) stays the reference's formula on torch tensors; its `post_stabilization`
(`engines.py:80-116`) solves the frictionless LCP with the HIP `LCPFunction`.
"""
import torch

from ..lcp.lcp import LCPFunction
from . import batched_world
from . import contacts as _contacts


class Engine:
    """Base class for stepping engine (`engines.py:11-14`)."""

    def solve_dynamics(self, world, dt):
        raise NotImplementedError


class HipPdipmEngine(Engine):
    """Engine that uses the MI355X primal-dual interior point LCP solver."""

    def __init__(self, max_iter=10):
        self.lcp_solver = LCPFunction
        self.cached_inverse = None
        self.max_iter = max_iter

    def _no_contact(self, world, u, Je, neq):
        # engines.py:35-49 (Cline eq. 2.41): [[M, -Je^T], [Je, 0]] x = u
        if neq > 0:
            P = torch.cat([torch.cat([world.M(), -Je.t()], dim=1),
                           torch.cat([Je, Je.new_zeros(neq, neq)], dim=1)])
        else:
            P = world.M()
        if self.cached_inverse is None:
            inv = torch.inverse(P)
            if world.static_inverse:
                self.cached_inverse = inv
        else:
            inv = self.cached_inverse
        return torch.matmul(inv, u)

    def solve_dynamics(self, world, dt):
        t = world.t
        Je = world.Je()
        neq = Je.size(0) if Je.ndimension() > 0 else 0
        f = world.apply_forces(t)
        u = torch.matmul(world.M(), world.get_v()) + dt * f            # engines.py:32
        if neq > 0:
            u = torch.cat([u, u.new_zeros(neq)])
        if not world.contacts:
            x = self._no_contact(world, u, Je, neq)
        else:
            Jc = world.Jc()
            v = torch.matmul(Jc, world.get_v()) * world.restitutions()  # engines.py:53
            M = world.M().unsqueeze(0)
            if neq > 0:
                b = Je.new_zeros(Je.size(0)).unsqueeze(0)
                Je = Je.unsqueeze(0)
            else:
                b = torch.tensor([])
                Je = torch.tensor([])
            Jc = Jc.unsqueeze(0)
            u = u[:world.M().size(0)].unsqueeze(0)
            v = v.unsqueeze(0)
            E = world.E().unsqueeze(0)
            mu = world.mu().unsqueeze(0)
            Jf = world.Jf().unsqueeze(0)
            nc, nf = Jc.size(1), Jf.size(1)
            G = torch.cat([Jc, Jf, Jf.new_zeros(1, nc, Jf.size(2))], dim=1)         # engines.py:67-68
            F = G.new_zeros(1, G.size(1), G.size(1))
            F[:, nc:nc + nf, nc + nf:] = E                                           # engines.py:70
            F[:, nc + nf:, :nc] = mu                                                 # engines.py:71
            F[:, nc + nf:, nc:nc + nf] = -E.transpose(1, 2)                          # engines.py:72-73
            h = torch.cat([v, v.new_zeros(1, nf + nc)], 1)                           # engines.py:74
            x = -self.lcp_solver(max_iter=self.max_iter, verbose=-1)(M, u, G, h, Je, b, F)
        new_v = x[:world.vec_len * len(world.bodies)].squeeze(0)
        return new_v

    def post_stabilization(self, world):
        # engines.py:80-116; the contact case is a frictionless LCP (G = Jc, F = 0)
        v = world.get_v()
        M = world.M()
        Je = world.Je()
        Jc = world.Jc() if world.contacts else None
        ge = torch.matmul(Je, v)
        u = torch.cat([Je.new_zeros(Je.size(1)), ge])
        if Jc is None:
            neq = Je.size(0) if Je.ndimension() > 0 else 0
            x = self._no_contact(world, u, Je, neq)
        else:
            gc = torch.matmul(Jc, v) + torch.matmul(Jc, v) * -world.restitutions()
            F = Jc.new_zeros(1, Jc.size(0), Jc.size(0))
            x = self.lcp_solver()(M.unsqueeze(0), u[:M.size(0)].unsqueeze(0), Jc.unsqueeze(0),
                                  gc.unsqueeze(0), Je.unsqueeze(0), u[M.size(0):].unsqueeze(0), F)
        return -x[:M.size(0)]


class HipFusedEngine(HipPdipmEngine):
    """Non-differentiable engine: the world's raw state goes to the device entry points of the batched path as a batch
    of one - `lcp_solve_dynamics_f32` (both branches of `engines.py:26-78`: a world without contacts takes the direct
    KKT solve inside the same kernel) and `lcp_post_stabilization_f32` (`engines.py:80-116`).  Joints are read
    through `world.Je()`, so pose-dependent ones work (the Jacobian is re-read every call)."""

    def __init__(self, max_iter=10, compute="f64"):
        super().__init__(max_iter=max_iter)
        self.compute = compute

    @staticmethod
    def _device_state(world):
        nb = len(world.bodies)
        dev = torch.device("cuda")
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        Je = world.Je()
        e = Je.size(0) if Je.ndimension() > 0 and Je.numel() > 0 else 0
        cs = world.contacts or []
        maxc = max(1, len(cs))
        cb = _contacts.ContactBuffers(1, nb, maxc, dev)
        if cs:
            st = lambda k: torch.stack([c[0][k].detach().reshape(2) for c in cs]).unsqueeze(0)
            cb.c_n, cb.c_p1, cb.c_p2 = f32(st(0)), f32(st(1)), f32(st(2))
            cb.c_i1 = torch.tensor([[int(c[1]) for c in cs]], dtype=torch.int32, device=dev)
            cb.c_i2 = torch.tensor([[int(c[2]) for c in cs]], dtype=torch.int32, device=dev)
        cb.count.fill_(len(cs))
        return dict(
            nb=nb, e=e, maxc=maxc, cb=cb,
            v=f32(world.get_v().reshape(1, nb, 3)), Mdiag=f32(torch.diagonal(world.M()).reshape(1, nb, 3)),
            rest=f32(torch.stack([b.restitution.reshape(()) for b in world.bodies]).unsqueeze(0)),
            fric=f32(torch.stack([b.fric_coeff.reshape(()) for b in world.bodies]).unsqueeze(0)),
            Je=f32(Je.unsqueeze(0)) if e else None)

    def solve_dynamics(self, world, dt):
        base = world.get_v()
        with torch.no_grad():
            d = self._device_state(world)
            f = world.apply_forces(world.t).detach().reshape(1, d["nb"], 3).to(device="cuda", dtype=torch.float32)
            out = batched_world.solve_dynamics(1, d["nb"], d["maxc"], d["e"], d["cb"].count, d["Mdiag"], d["v"],
                                               f.contiguous(), d["rest"], d["fric"], d["cb"], d["Je"], float(dt),
                                               max_iter=self.max_iter, compute=self.compute)
            return out["v_new"].reshape(-1).to(device=base.device, dtype=base.dtype)

    def post_stabilization(self, world):
        base = world.get_v()
        with torch.no_grad():
            d = self._device_state(world)
            out = batched_world.post_stabilization(1, d["nb"], d["maxc"], d["e"], d["cb"].count, d["Mdiag"], d["v"],
                                                   d["rest"], d["cb"], d["Je"], compute=self.compute)
            return out["dp"].reshape(-1).to(device=base.device, dtype=base.dtype)

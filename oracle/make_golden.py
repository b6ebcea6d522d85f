"""Generate the committed golden fixtures under tests/golden/ by running the
UNMODIFIED reference (through oracle/ref_shim.py) in this container.

TEST INFRASTRUCTURE ONLY.  Usage (needs /root/reference, so only here):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference has no golden vectors of its own, so parity is
pinned on outputs of the reference itself:

* scenes are built with the reference's own `World`/`Rect`/`Circle` classes, in the
  same way as its demos/tests (`demos/demo.py`, `tests/test_demos.py:102-177`);
* every recorded simulation step stores the world state entering
  `PdipmEngine.solve_dynamics` (`physics/engines.py:26`), the dense LCP handed to
  `LCPFunction` (`engines.py:76`), the reference solution (x, lams, slacks, nus),
  the returned `new_v`, and the 7 gradients of `LCPFunction.backward`
  (`lcp/lcp.py:37-64`) for a seeded cotangent.

Everything is float64 (the reference default, `physics/utils.py:34`).
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _scenes():
    from lcp_physics.physics.bodies import Circle, Rect
    from lcp_physics.physics.constraints import TotalConstraint, XConstraint, YConstraint
    from lcp_physics.physics.forces import ExternalForce, Gravity, down_force

    def ball_floor():
        c = Circle([500, 300], 30)
        c.add_force(Gravity(g=100))
        r = Rect([500, 400], [900, 10])
        return [c, r], [TotalConstraint(r)], 400, 1

    def stack(nbox, jitter):
        def make():
            bodies, joints = [], []
            floor = Rect([500, 500], [900, 10])
            bodies.append(floor)
            joints.append(TotalConstraint(floor))
            y = 495.0
            for i in range(nbox):
                y -= 30.0
                b = Rect([500 + jitter[i], y], [60, 60], mass=1.0 + 0.25 * i,
                         fric_coeff=0.5 + 0.1 * i, restitution=0.1 * (i + 1))
                y -= 30.0
                b.add_force(Gravity(g=100))
                bodies.append(b)
            return bodies, joints, 40, 6
        return make

    def slide():
        # tests/test_demos.py:102-137
        bodies, joints = [], []
        r = Rect([500, 300], [900, 10])
        r.v[0] = math.pi / 32
        r.move(1)
        r.v[0] = 0.
        bodies.append(r)
        joints.append(TotalConstraint(r))
        r = Rect([100, 100], [60, 60])
        r.move(1)
        bodies.append(r)
        r.add_force(ExternalForce(down_force, multiplier=100))
        return bodies, joints, 240, 25

    def fric():
        # tests/test_demos.py:139-177 (without the free-flying "clock" body)
        restitution, fric_coeff = 0.75, 1

        def timed_force(t):
            return ExternalForce.RIGHT if 1 < t < 2 else ExternalForce.ZEROS

        bodies, joints = [], []
        r = Rect([400, 400], [900, 10], restitution=restitution, fric_coeff=fric_coeff)
        bodies.append(r)
        r.add_force(ExternalForce(timed_force, multiplier=100))
        r.add_force(ExternalForce(down_force, multiplier=100))
        c = Circle([200, 364], 30, restitution=restitution, fric_coeff=fric_coeff)
        bodies.append(c)
        c.add_force(ExternalForce(down_force, multiplier=100))
        for x in (50, 800):
            c = Circle([x, 436], 30, restitution=restitution, fric_coeff=fric_coeff)
            bodies.append(c)
            joints.append(XConstraint(c))
            joints.append(YConstraint(c))
        return bodies, joints, 90, 8

    return {
        "ball_floor": ball_floor,
        "stack2": stack(2, [3.0, -4.0]),
        "stack4": stack(4, [3.0, -4.0, 5.0, -2.0]),
        "slide": slide,
        "fric": fric,
    }


def record_scene(name, make, mods):
    from lcp_physics.physics.world import World
    torch.manual_seed(0)
    bodies, joints, n_steps, every = make()
    world = World(bodies, joints)
    engine = world.engine
    rec = []
    state = {}

    lcp_cls = mods["engines"].LCPFunction

    class Spy(lcp_cls):
        def __call__(self, *args):
            args = [a.detach().clone() for a in args]
            ins = [a.clone().requires_grad_(True) if a.numel() else a for a in args]
            out = lcp_cls.__call__(self, *ins)
            g = torch.Generator().manual_seed(1000 + len(rec))
            cot = torch.randn(out.shape, generator=g, dtype=out.dtype)
            diff = [a for a in ins if a.numel()]
            grads = torch.autograd.grad(out, diff, cot, allow_unused=True)
            gi = iter(grads)
            full = [next(gi) if a.numel() else None for a in ins]
            state["lcp"] = dict(args=args, x=out.detach().clone(), lams=self.lams.clone(),
                                slacks=self.slacks.clone(),
                                nus=None if self.nus is None else self.nus.clone(),
                                cot=cot, grads=full)
            return out.detach()

    engine.lcp_solver = Spy
    orig_solve = engine.solve_dynamics

    def solve_dynamics(w, dt):
        state.clear()
        pre = dict(
            t=w.t, dt=dt,
            p=torch.stack([b.p for b in w.bodies]).clone(),
            v=w.get_v().clone().reshape(len(w.bodies), 3),
            Mdiag=torch.diagonal(w.M()).clone().reshape(len(w.bodies), 3),
            rest=torch.stack([b.restitution.reshape(()) for b in w.bodies]),
            fric=torch.stack([b.fric_coeff.reshape(()) for b in w.bodies]),
            f=w.apply_forces(w.t).clone().reshape(len(w.bodies), 3),
            Je=w.Je().clone(),
            contacts=[(c[0][0].clone(), c[0][1].clone(), c[0][2].clone(),
                       torch.as_tensor(c[0][3]).reshape(()).clone(), c[1], c[2])
                      for c in w.contacts],
        )
        new_v = orig_solve(w, dt)
        if "lcp" in state:
            rec.append((pre, state["lcp"], new_v.detach().clone()))
        return new_v

    engine.solve_dynamics = solve_dynamics
    for _ in range(n_steps):
        world.step()
    picked = rec[::every][:8]
    print("%-10s lcp steps recorded %4d, kept %d, shapes %s" % (
        name, len(rec), len(picked),
        sorted({tuple(r[1]["args"][2].shape[1:]) for r in picked})))
    return picked


def save_scene(name, picked):
    out = {}
    for k, (pre, lcp, new_v) in enumerate(picked):
        pfx = "s%d_" % k
        nc = len(pre["contacts"])
        out[pfx + "dt"] = np.float64(pre["dt"])
        for key in ("p", "v", "Mdiag", "rest", "fric", "f", "Je"):
            out[pfx + key] = pre[key].numpy()
        out[pfx + "c_n"] = torch.stack([c[0] for c in pre["contacts"]]).numpy()
        out[pfx + "c_p1"] = torch.stack([c[1] for c in pre["contacts"]]).numpy()
        out[pfx + "c_p2"] = torch.stack([c[2] for c in pre["contacts"]]).numpy()
        out[pfx + "c_pen"] = torch.stack([c[3] for c in pre["contacts"]]).numpy()
        out[pfx + "c_i1"] = np.array([c[4] for c in pre["contacts"]], dtype=np.int32)
        out[pfx + "c_i2"] = np.array([c[5] for c in pre["contacts"]], dtype=np.int32)
        for nm, a in zip("QpGhAbF", lcp["args"]):
            out[pfx + "in_" + nm] = a.numpy()
        out[pfx + "x"] = lcp["x"].numpy()
        out[pfx + "lams"] = lcp["lams"].numpy()
        out[pfx + "slacks"] = lcp["slacks"].numpy()
        if lcp["nus"] is not None:
            out[pfx + "nus"] = lcp["nus"].numpy()
        out[pfx + "cot"] = lcp["cot"].numpy()
        for nm, g in zip("QpGhAbF", lcp["grads"]):
            if g is not None:
                out[pfx + "grad_" + nm] = g.numpy()
        out[pfx + "new_v"] = new_v.numpy()
        assert nc == out[pfx + "in_G"].shape[1] // 4
    out["n_steps"] = np.int64(len(picked))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("   wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def main():
    mods = ref_shim.load_reference()
    os.makedirs(OUT, exist_ok=True)
    for name, make in _scenes().items():
        save_scene(name, record_scene(name, make, mods))


if __name__ == "__main__":
    main()

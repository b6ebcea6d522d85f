"""Generate tests/golden/world_traj.npz: trajectories of the UNMODIFIED reference `World` (physics/world.py,
through oracle/ref_shim.py) for small scenes that exercise contact creation, the penetration check with dt
halving (world.py:88-101), the no-contact branch of the engine (engines.py:36-50) and - the `*_poststab` records -
post-stabilisation (world.py:109-121, engines.py:80-116).
TEST INFRASTRUCTURE ONLY; needs /root/reference.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_world.py
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _scenes():
    from lcp_physics.physics.bodies import Circle, Rect
    from lcp_physics.physics.constraints import TotalConstraint
    from lcp_physics.physics.forces import Gravity

    def floor():
        r = Rect([500, 500], [900, 10])
        return r, TotalConstraint(r)

    def ball_floor():
        fl, j = floor()
        c = Circle([500, 420], 30, restitution=0.6)
        c.add_force(Gravity(g=100))
        return [fl, c], [j], 45, True

    def stack3():
        fl, j = floor()
        bodies = [fl]
        y = 495.0
        for i, dx in enumerate((0.0, 7.0, -5.0)):
            y -= 20.0
            b = Rect([500 + dx, y - 0.5 - 2.0 * i], [40, 40], mass=1.0 + 0.5 * i, fric_coeff=0.4 + 0.1 * i,
                     restitution=0.2 + 0.1 * i)
            y -= 20.0
            b.add_force(Gravity(g=100))
            bodies.append(b)
        return bodies, [j], 30, True

    def mixed(strict):
        def make():
            fl, j = floor()
            a = Rect([470, 470], [50, 50], vel=[0, 30, 0], fric_coeff=0.3)
            b = Circle([560, 440], 20, vel=[0, -60, 0], restitution=0.3)
            c = Rect([0.3, 520, 380], [30, 20], mass=0.5)
            for x in (a, b, c):
                x.add_force(Gravity(g=100))
            return [fl, a, b, c], [j], 40, strict
        return make

    def tumble():
        fl, j = floor()
        a = Rect([0.4, 500, 440], [60, 30], vel=[1.0, 20, 0], restitution=0.3)
        a.add_force(Gravity(g=100))
        return [fl, a], [j], 45, True

    def chain():
        # tests/test_demos.py:76-100 (testChain), five links: a pinned link, revolute `Joint`s whose Jacobian follows the pose
        # (constraints.py:13-53), a constant pull on the last link and a projectile pushed by a force that stops at t = 0.1
        # (forces.py:14-18 hor_impulse)
        from lcp_physics.physics.constraints import Joint, XConstraint, YConstraint
        from lcp_physics.physics.forces import ExternalForce, down_force, hor_impulse
        bodies, joints = [], []
        r = Rect([300, 50], [20, 60])
        bodies.append(r)
        joints.append(XConstraint(r))
        joints.append(YConstraint(r))
        for i in range(1, 5):
            r = Rect([300, 50 + 50 * i], [20, 60])
            bodies.append(r)
            joints.append(Joint(bodies[-1], bodies[-2], [300, 25 + 50 * i]))
            bodies[-1].add_no_contact(bodies[-2])
        bodies[-1].add_force(ExternalForce(down_force, multiplier=100))
        c = Circle([183.7, 240], 20, restitution=1)          # (not a multiple of the step: no exactly-touching pose)
        bodies.append(c)
        c.add_force(ExternalForce(hor_impulse, multiplier=1000))
        return bodies, joints, 60, True

    def welded():
        # two boxes welded by a FixedJoint (constraints.py:56-92) tumbling onto the floor
        from lcp_physics.physics.constraints import FixedJoint
        fl, j = floor()
        a = Rect([0.3, 470, 430], [50, 30], vel=[0.5, 10, 0])
        b = Rect([0.3, 520, 445.5], [50, 30])
        for x in (a, b):
            x.add_force(Gravity(g=100))
        a.add_no_contact(b)
        return [fl, a, b], [j, FixedJoint(a, b)], 40, True

    return dict(ball_floor=ball_floor, stack3=stack3, mixed=mixed(True), mixed_nonstrict=mixed(False), tumble=tumble,
                chain=chain, welded=welded,
                # the same scenes with post-stabilisation switched on (world.py:109-121, engines.py:80-116)
                stack3_poststab=stack3, mixed_poststab=mixed(True), tumble_poststab=tumble)


def record(name, make):
    from lcp_physics.physics.bodies import Circle
    from lcp_physics.physics.world import World
    random.seed(0)
    bodies, joints, nsteps, strict = make()
    post_stab = name.endswith("_poststab")
    world = World(bodies, joints, dt=1.0 / 30, strict_no_penetration=strict, post_stab=post_stab)
    nb = len(bodies)
    kind = np.array([0 if isinstance(b, Circle) else 1 for b in bodies])
    size = np.array([[float(b.rad), 0.0] if isinstance(b, Circle) else b.dims.numpy().tolist() for b in bodies])
    rec = dict(kind=kind, size=size, dt=np.float64(world.dt), strict=np.int64(strict), post_stab=np.int64(post_stab),
               eps=np.float64(float(world.eps)),
               tol=np.float64(float(world.tol)),
               Mdiag=torch.diagonal(world.M()).reshape(nb, 3).numpy().copy(),
               f=world.apply_forces(0).reshape(nb, 3).numpy().copy(),
               rest=np.array([float(b.restitution) for b in bodies]),
               fric=np.array([float(b.fric_coeff) for b in bodies]),
               Je=world.Je().numpy().copy())
    # joints as a batched world needs them (constraints.py): type, bodies, and for `Joint` the polar coordinates of the anchor
    from lcp_physics.physics import constraints as C_
    jt = {C_.Joint: 1, C_.FixedJoint: 2, C_.XConstraint: 3, C_.YConstraint: 4, C_.RotConstraint: 5, C_.TotalConstraint: 6}
    rec.update(jtype=np.array([jt[type(j[0])] for j in world.joints]),
               jb1=np.array([j[1] for j in world.joints]), jb2=np.array([-1 if j[2] is None else j[2] for j in world.joints]),
               jr1=np.array([float(j[0].r1) if isinstance(j[0], C_.Joint) else 0.0 for j in world.joints]),
               jrot1=np.array([float(j[0].rot1) if isinstance(j[0], C_.Joint) else 0.0 for j in world.joints]),
               no_contact=np.array([[i, k] for i, b in enumerate(bodies) for k, o in enumerate(bodies) if o.geom in b.geom.no_contact]).reshape(-1, 2))
    P, V, T, NC, DP, PMID, FT, JE = [], [], [], [], [], [], [], []
    if post_stab:
        # per step: the engine's post_stabilization output and the pose it was computed at (the pose before the
        # post-stabilisation move) - lets the test pin that solve step by step on identical inputs.  The engine INSTANCE
        # is wrapped; no reference file is touched.
        inner = world.engine.post_stabilization
        solver_cls = world.engine.lcp_solver
        LCPS = []                    # the frictionless LCPs the engine handed to the solver: (step, Jc, gc, ge, x)
        state = dict(on=False, step=0)

        class SolverSpy:
            def __init__(self, *a, **k):
                self.f = solver_cls(*a, **k)

            def __call__(self, Q, p_, G_, h_, A_, b_, F_):
                x = self.f(Q, p_, G_, h_, A_, b_, F_)
                if state["on"]:
                    assert float(p_.abs().max()) == 0.0 and float(F_.abs().max()) == 0.0      # engines.py:84,110
                    LCPS.append((state["step"], G_[0].numpy().copy(), h_[0].numpy().copy(), b_[0].numpy().copy(),
                                 x.detach()[0].numpy().copy()))
                return x
        world.engine.lcp_solver = SolverSpy

        def spy(wd):
            state["on"] = True
            out = inner(wd)
            state["on"] = False
            state["step"] += 1
            PMID.append(torch.stack([b.p for b in bodies]).numpy().copy())
            DP.append(out.detach().reshape(nb, 3).numpy().copy())
            return out
        world.engine.post_stabilization = spy
    snap = lambda: (torch.stack([b.p for b in bodies]).numpy().copy(), world.get_v().reshape(nb, 3).numpy().copy())
    p, v = snap()
    P.append(p); V.append(v); T.append(float(world.t)); NC.append(len(world.contacts))
    for _ in range(nsteps):
        FT.append(world.apply_forces(world.t).reshape(nb, 3).detach().numpy().copy())     # the force the step's solve sees
        JE.append(world.Je().detach().numpy().copy())                                     # ... and its joint Jacobian
        world.step()
        p, v = snap()
        P.append(p); V.append(v); T.append(float(world.t)); NC.append(len(world.contacts))
    JE.append(world.Je().detach().numpy().copy())
    rec.update(p=np.stack(P), v=np.stack(V), t=np.array(T), ncontacts=np.array(NC), f_t=np.stack(FT), Je_t=np.stack(JE))
    if post_stab:
        rec.update(dp=np.stack(DP), p_mid=np.stack(PMID))
        cap = max(1, max(l[1].shape[0] for l in LCPS))
        nl = len(LCPS)
        Jc = np.zeros((nl, cap, 3 * nb)); gc = np.zeros((nl, cap)); ge = np.zeros((nl, LCPS[0][3].shape[0])); xs = np.zeros((nl, 3 * nb))
        for i, (_, G_, h_, b_, x_) in enumerate(LCPS):
            Jc[i, :G_.shape[0]] = G_; gc[i, :h_.shape[0]] = h_; ge[i] = b_; xs[i] = x_[:3 * nb]
        rec.update(ps_step=np.array([l[0] for l in LCPS]), ps_nc=np.array([l[1].shape[0] for l in LCPS]), ps_Jc=Jc, ps_gc=gc,
                   ps_ge=ge, ps_x=xs)
    return rec


def main():
    ref_shim.load_reference()
    torch.set_default_dtype(torch.float64)
    flat = {}
    names = []
    for name, make in _scenes().items():
        rec = record(name, make)
        names.append(name)
        for k, v in rec.items():
            flat["%s__%s" % (name, k)] = v
        dts = np.diff(rec["t"])
        print(name, "steps", len(dts), "contacts", rec["ncontacts"].tolist(), "\n   halved steps:", int((dts < rec["dt"] * 0.99).sum()),
              "min dt", dts.min())
    flat["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "world_traj.npz"), **flat)


if __name__ == "__main__":
    main()

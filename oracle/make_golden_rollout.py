"""Generate tests/golden/rollout_grad.npz: gradients the UNMODIFIED reference obtains by back-propagating through many
`World.step`s (what `demos/grad_demo.py:19-83` does: `dist = (target.pos - c.pos).norm(); dist.backward()` after a
roll-out, gradient with respect to a force applied for the first 0.1 s; `experiments/inference.py:26-89` has the same
shape).  TEST INFRASTRUCTURE ONLY; needs /root/reference.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_rollout.py

Scene = grad_demo's `make_world` (three circles of radius 30: a target that collides with nothing, a pushed ball c1 and the
ball c2 it has to send towards the target), with geometry / force magnitudes chosen so that the collision happens inside
a 36-step roll-out.  For each of several initial forces the fixture stores the final state, the loss and
d(loss)/d(initial_force) from the reference's autograd - through `PdipmEngine.solve_dynamics`, `LCPFunction.backward`,
`DiffContactHandler` (circle / circle: physics/contacts.py:67-80) and `Body.move` (bodies.py:80-96) - plus what a batched
world needs to rebuild the scene.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
NSTEPS = 36
T_PUSH = 0.1
MULT = 100.0
FORCES = [[0.0, 30.0, 8.0], [0.0, 34.0, 9.5], [0.0, 27.0, 6.0], [0.5, 31.0, 8.8], [0.0, 36.0, 11.0], [-0.4, 29.0, 7.1],
          [0.0, 32.5, 10.2], [0.2, 28.0, 8.0]]


def make_world(force_fn):
    from lcp_physics.physics.bodies import Circle
    from lcp_physics.physics.forces import ExternalForce
    from lcp_physics.physics.world import World
    target = Circle([500, 300], 30)
    c1 = Circle([250, 210], 30)
    c1.add_force(ExternalForce(force_fn, multiplier=MULT))
    c1.add_no_contact(target)
    c2 = Circle([345, 236], 30, restitution=0.4, fric_coeff=0.3)
    c2.add_no_contact(target)
    world = World([target, c1, c2], [], dt=1.0 / 30)
    return world, c2, target


def run(force0):
    from lcp_physics.physics.forces import ExternalForce
    f0 = torch.tensor(force0, dtype=torch.float64, requires_grad=True)
    fn = lambda t: f0 if t < T_PUSH else ExternalForce.ZEROS
    world, c, target = make_world(fn)
    nb = len(world.bodies)
    rec = dict(Mdiag=torch.diagonal(world.M()).reshape(nb, 3).detach().numpy().copy(),
               rest=np.array([float(b.restitution) for b in world.bodies]),
               fric=np.array([float(b.fric_coeff) for b in world.bodies]),
               rad=np.array([float(b.rad) for b in world.bodies]),
               p0=torch.stack([b.p for b in world.bodies]).detach().numpy().copy(),
               v0=world.get_v().reshape(nb, 3).detach().numpy().copy())
    ncs, ts = [], []
    for _ in range(NSTEPS):
        world.step()
        ncs.append(len(world.contacts)); ts.append(float(world.t))
    dist = (target.pos - c.pos).norm()
    dist.backward()
    rec.update(p_final=torch.stack([b.p for b in world.bodies]).detach().numpy().copy(),
               v_final=world.get_v().reshape(nb, 3).detach().numpy().copy(),
               loss=np.float64(float(dist)), grad=f0.grad.numpy().copy(), ncontacts=np.array(ncs), t=np.array(ts))
    return rec


# ---- second scene: contacts that involve hulls (circle / hull by GJK, hull / hull by SAT + clipping) ------------------------
# a floor (TotalConstraint), a ball and a box, both under gravity, both pushed for the first 0.1 s by learnable forces; the ball
# lands on the floor, rolls into the box, the box slides and tips.  loss = |ball - box| after the roll-out.
# (In three of the scenes the box carries no torque and stays flat, |rot| ~ 1e-17: the steps in which its rotation increment is
#  exactly zero contribute no vertex-rotation path to the reference's gradient - bodies.py:199-202 skips `rotate_verts` then -
#  which shows up in d(loss)/d(box torque); `box_rot_max` is recorded to tell those scenes apart.)
H_NSTEPS = 40
H_FORCES = [([0.0, 6.0, 0.0], [0.0, -2.0, 0.0]), ([0.0, 8.0, 1.0], [0.3, -3.0, 0.0]), ([0.0, 5.0, -1.0], [0.0, -1.0, 0.5]),
            ([0.2, 7.0, 0.5], [-0.2, -2.5, 0.0]), ([0.0, 9.0, 0.0], [0.0, -4.0, 1.0]), ([-0.1, 6.5, 2.0], [0.1, -1.5, -0.5])]


def make_world_hulls(f_ball, f_box):
    from lcp_physics.physics.bodies import Circle, Rect
    from lcp_physics.physics.constraints import TotalConstraint
    from lcp_physics.physics.forces import ExternalForce, Gravity
    from lcp_physics.physics.world import World
    floor = Rect([500, 500], [900, 10])
    ball = Circle([380, 468], 20, restitution=0.3, fric_coeff=0.6)
    box = Rect([470, 474.5], [40, 40], restitution=0.2, fric_coeff=0.4)
    for b, fn in ((ball, f_ball), (box, f_box)):
        b.add_force(Gravity(g=100))
        b.add_force(ExternalForce(fn, multiplier=MULT))
    world = World([floor, ball, box], [TotalConstraint(floor)], dt=1.0 / 30)
    return world, ball, box


def run_hulls(fb, fx):
    from lcp_physics.physics.bodies import Circle
    from lcp_physics.physics.forces import ExternalForce
    f1 = torch.tensor(fb, dtype=torch.float64, requires_grad=True)
    f2 = torch.tensor(fx, dtype=torch.float64, requires_grad=True)
    world, ball, box = make_world_hulls(lambda t: f1 if t < T_PUSH else ExternalForce.ZEROS,
                                        lambda t: f2 if t < T_PUSH else ExternalForce.ZEROS)
    nb = len(world.bodies)
    rec = dict(Mdiag=torch.diagonal(world.M()).reshape(nb, 3).detach().numpy().copy(),
               rest=np.array([float(b.restitution) for b in world.bodies]),
               fric=np.array([float(b.fric_coeff) for b in world.bodies]),
               kind=np.array([0 if isinstance(b, Circle) else 1 for b in world.bodies]),
               size=np.array([[float(b.rad), 0.0] if isinstance(b, Circle) else b.dims.numpy().tolist() for b in world.bodies]),
               gravity=np.stack([np.zeros(3)] + [np.array([0.0, 0.0, 100.0 * float(b.mass)]) for b in world.bodies[1:]]),
               Je=world.Je().detach().numpy().copy(),
               p0=torch.stack([b.p for b in world.bodies]).detach().numpy().copy(),
               v0=world.get_v().reshape(nb, 3).detach().numpy().copy())
    ncs, ts, rot_max = [], [], 0.0
    for _ in range(H_NSTEPS):
        world.step()
        ncs.append(len(world.contacts)); ts.append(float(world.t))
        rot_max = max(rot_max, abs(float(box.p[0])))
    dist = (ball.pos - box.pos).norm()
    dist.backward()
    rec.update(p_final=torch.stack([b.p for b in world.bodies]).detach().numpy().copy(), loss=np.float64(float(dist)),
               grad_ball=f1.grad.numpy().copy(), grad_box=f2.grad.numpy().copy(), ncontacts=np.array(ncs), t=np.array(ts), box_rot_max=np.float64(rot_max))
    return rec


# ---- third and fourth scenes: joints whose Jacobian depends on the pose (Joint.J / FixedJoint.J, constraints.py:26-85) ------
# "j": a double pendulum (bob hinged to the world, a link hinged to the bob: 4 equality rows) that swings into a free ball;
# "k": a dumbbell (two discs welded by a FixedJoint: 3 rows) pushed, spinning, into a free ball.  Learnable forces push the
# first body and the ball for the first 0.1 s; loss = |ball - second body| after the roll-out.  The reference differentiates
# through pos1 = r1 (cos rot1, sin rot1) with rot1 += body1.v[0] dt and pos2 = body1.pos + pos1 - body2.pos - the path
# lcp_step_backward_je_f32's dL/dJe feeds.  (Scenes with at most 4 equality rows: the fused backward's limit.)
J_NSTEPS = 36
J_FORCES = [([0.0, 40.0, 0.0], [0.0, 0.0, 0.0]), ([0.0, 55.0, 5.0], [0.0, -3.0, 1.0]), ([2.0, 35.0, -4.0], [0.5, 2.0, 0.0]),
            ([0.0, 48.0, 10.0], [-0.5, -1.0, -2.0]), ([-1.0, 60.0, 0.0], [0.0, 1.0, 3.0]), ([0.0, 30.0, 6.0], [0.2, -2.0, 0.0])]
K_FORCES = [([0.0, 30.0, 0.0], [0.0, 0.0, 0.0]), ([3.0, 36.0, 4.0], [0.0, -2.0, 1.0]), ([-2.0, 28.0, -3.0], [0.3, 1.0, 0.0]),
            ([5.0, 33.0, 6.0], [-0.3, -1.0, -1.0]), ([1.0, 40.0, -5.0], [0.0, 0.5, 2.0]), ([-4.0, 26.0, 2.0], [0.1, -1.5, 0.0])]


def make_world_pendulum(f_first, f_ball, pos0=None):
    from lcp_physics.physics.bodies import Circle
    from lcp_physics.physics.constraints import Joint
    from lcp_physics.physics.forces import ExternalForce, Gravity
    from lcp_physics.physics.world import World
    pos0 = pos0 or ([300, 300], [300, 372], [390, 300])        # (tensors that require grad in the "a_" scenes: Body.__init__ keeps them)
    bob = Circle(pos0[0], 20, restitution=0.3, fric_coeff=0.4)
    link = Circle(pos0[1], 18, restitution=0.3, fric_coeff=0.4)
    ball = Circle(pos0[2], 22, restitution=0.5, fric_coeff=0.3)
    bob.add_no_contact(link)
    for b in (bob, link):
        b.add_force(Gravity(g=100))
    bob.add_force(ExternalForce(f_first, multiplier=MULT))
    ball.add_force(ExternalForce(f_ball, multiplier=MULT))
    world = World([bob, link, ball], [Joint(bob, None, [300, 220]), Joint(bob, link, [300, 330])], dt=1.0 / 30)
    return world, [[0, 1]], [0, 1]


def make_world_dumbbell(f_first, f_ball):
    from lcp_physics.physics.bodies import Circle
    from lcp_physics.physics.constraints import FixedJoint
    from lcp_physics.physics.forces import ExternalForce
    from lcp_physics.physics.world import World
    a = Circle([300, 300], 20, restitution=0.4, fric_coeff=0.5)
    b = Circle([352, 300], 16, restitution=0.4, fric_coeff=0.5)
    ball = Circle([430, 318], 22, restitution=0.5, fric_coeff=0.3)
    a.add_no_contact(b)
    a.add_force(ExternalForce(f_first, multiplier=MULT))
    ball.add_force(ExternalForce(f_ball, multiplier=MULT))
    world = World([a, b, ball], [FixedJoint(a, b)], dt=1.0 / 30)
    return world, [[0, 1]], []


def run_joints(make, fb, fx, pos0=None):
    """`pos0`: initial positions [nb][2] that REQUIRE GRAD ("a_" scenes) - the reference's `Joint.__init__` takes the anchor's polar
    coordinates from `pos - body1.pos` (constraints.py:21-23), so d(loss)/d(initial position) has a path through (r1, rot1) and
    every later `Joint.J()`; recorded as grad_p0 [nb,2]."""
    from lcp_physics.physics import constraints as C_
    from lcp_physics.physics.forces import ExternalForce
    f1 = torch.tensor(fb, dtype=torch.float64, requires_grad=True)
    f2 = torch.tensor(fx, dtype=torch.float64, requires_grad=True)
    leaves = None if pos0 is None else [torch.tensor(q, dtype=torch.float64, requires_grad=True) for q in pos0]
    world, no_contact, heavy = make(lambda t: f1 if t < T_PUSH else ExternalForce.ZEROS,
                                    lambda t: f2 if t < T_PUSH else ExternalForce.ZEROS, *([] if leaves is None else [leaves]))
    second, ball = world.bodies[1], world.bodies[2]
    nb = len(world.bodies)
    jt = {C_.Joint: 1, C_.FixedJoint: 2, C_.XConstraint: 3, C_.YConstraint: 4, C_.RotConstraint: 5, C_.TotalConstraint: 6}
    rec = dict(Mdiag=torch.diagonal(world.M()).reshape(nb, 3).detach().numpy().copy(),
               rest=np.array([float(b.restitution) for b in world.bodies]),
               fric=np.array([float(b.fric_coeff) for b in world.bodies]),
               rad=np.array([float(b.rad) for b in world.bodies]),
               gravity=np.stack([np.array([0.0, 0.0, 100.0 * float(b.mass)]) if i in heavy else np.zeros(3) for i, b in enumerate(world.bodies)]),
               jtype=np.array([jt[type(j[0])] for j in world.joints]), jb1=np.array([j[1] for j in world.joints]),
               jb2=np.array([-1 if j[2] is None else j[2] for j in world.joints]),
               jr1=np.array([float(j[0].r1) if isinstance(j[0], C_.Joint) else 0.0 for j in world.joints]),
               jrot1=np.array([float(j[0].rot1) if isinstance(j[0], C_.Joint) else 0.0 for j in world.joints]),
               Je=world.Je().detach().numpy().copy(), no_contact=np.array(no_contact),
               p0=torch.stack([b.p for b in world.bodies]).detach().numpy().copy(),
               v0=world.get_v().reshape(nb, 3).detach().numpy().copy())
    ncs, ts = [], []
    for _ in range(J_NSTEPS):
        world.step()
        ncs.append(len(world.contacts)); ts.append(float(world.t))
    dist = (ball.pos - second.pos).norm()
    dist.backward()
    rec.update(p_final=torch.stack([b.p for b in world.bodies]).detach().numpy().copy(), loss=np.float64(float(dist)),
               grad_first=f1.grad.numpy().copy(), grad_ball=f2.grad.numpy().copy(), ncontacts=np.array(ncs), t=np.array(ts))
    if leaves is not None:
        rec["grad_p0"] = np.stack([q.grad.numpy().copy() for q in leaves])
    return rec


# ---- fifth scene: experiments/inference.py:26-89 in small - a chain of four links (revolute joints, 8 equality rows) hit by a
# projectile, World(post_stab=True); the learnable parameters are the links' mass (it enters the inertia, the mass and gravity,
# inference.py:104-113) and the projectile's push; loss = mean squared distance of the final poses from fixed targets.  The
# reference differentiates through solve_dynamics, the joints, the contacts AND post_stabilization (engines.py:80-116).
C_NSTEPS = 30
C_LINKS = 4
C_PARAMS = [(0.7, [0.0, 1.0, 0.0]), (1.2, [0.0, 1.2, 0.1]), (0.45, [0.0, 0.9, -0.1]), (2.0, [0.1, 1.1, 0.0]), (0.9, [0.0, 1.4, 0.05]),
            (1.5, [-0.1, 0.8, 0.0])]


def make_world_chain(mass, f_proj, links=C_LINKS):
    from lcp_physics.physics.bodies import Circle, Rect
    from lcp_physics.physics.constraints import Joint
    from lcp_physics.physics.forces import ExternalForce, Gravity
    from lcp_physics.physics.world import World
    bodies, joints = [], []
    r = Rect([300, 50], [20, 60], mass=mass)
    bodies.append(r)
    joints.append(Joint(r, None, [300, 30]))
    for i in range(1, links):
        r = Rect([300, 50 + 50 * i], [20, 60], mass=mass)
        r.add_force(Gravity(g=100))
        bodies.append(r)
        joints.append(Joint(bodies[-1], bodies[-2], [300, 25 + 50 * i]))
        bodies[-1].add_no_contact(bodies[-2])
    c = Circle([231.7, float(bodies[-1].pos[1]) + 3.3], 20, restitution=1.0)   # (off the grid: x = 230 + 150 t touches the link EXACTLY at a step)
    bodies.append(c)
    c.add_force(ExternalForce(f_proj, multiplier=1500))
    return World(bodies, joints, dt=1.0 / 30, post_stab=True)


def run_chain(mass0, force0, links=C_LINKS, nsteps=C_NSTEPS):
    from lcp_physics.physics import constraints as C_
    from lcp_physics.physics.bodies import Circle
    from lcp_physics.physics.forces import ExternalForce
    mass = torch.tensor(mass0, dtype=torch.float64, requires_grad=True)
    f0 = torch.tensor(force0, dtype=torch.float64, requires_grad=True)
    world = make_world_chain(mass, lambda t: f0 if t < T_PUSH else ExternalForce.ZEROS, links)
    nb = len(world.bodies)
    jt = {C_.Joint: 1, C_.FixedJoint: 2, C_.XConstraint: 3, C_.YConstraint: 4, C_.RotConstraint: 5, C_.TotalConstraint: 6}
    Md = torch.diagonal(world.M()).reshape(nb, 3).detach().numpy().copy()
    rec = dict(Mdiag=Md, Mdiag_per_mass=np.concatenate([Md[:links] / mass0, np.zeros((1, 3))]),
               gravity_per_mass=np.array([[0.0, 0.0, 0.0]] + [[0.0, 0.0, 100.0]] * (links - 1) + [[0.0, 0.0, 0.0]]),
               rest=np.array([float(b.restitution) for b in world.bodies]), fric=np.array([float(b.fric_coeff) for b in world.bodies]),
               kind=np.array([0 if isinstance(b, Circle) else 1 for b in world.bodies]),
               size=np.array([[float(b.rad), 0.0] if isinstance(b, Circle) else b.dims.numpy().tolist() for b in world.bodies]),
               jtype=np.array([jt[type(j[0])] for j in world.joints]), jb1=np.array([j[1] for j in world.joints]),
               jb2=np.array([-1 if j[2] is None else j[2] for j in world.joints]),
               jr1=np.array([float(j[0].r1) for j in world.joints]), jrot1=np.array([float(j[0].rot1) for j in world.joints]),
               no_contact=np.array([[i, i - 1] for i in range(1, links)]),
               p0=torch.stack([b.p for b in world.bodies]).detach().numpy().copy(),
               v0=world.get_v().reshape(nb, 3).detach().numpy().copy())
    ncs, ts = [], []
    for _ in range(nsteps):
        world.step()
        ncs.append(len(world.contacts)); ts.append(float(world.t))
    pf = torch.stack([b.p for b in world.bodies])
    target = torch.tensor(rec["p0"]) + torch.tensor([0.3, 25.0, -10.0])
    loss = ((pf - target) ** 2).mean()
    loss.backward()
    rec.update(p_final=pf.detach().numpy().copy(), loss=np.float64(float(loss)), grad_mass=mass.grad.numpy().copy(),
               grad_force=f0.grad.numpy().copy(), ncontacts=np.array(ncs), t=np.array(ts), target=target.numpy().copy())
    return rec


# the full size of experiments/inference.py: ten links, 20 equality rows, 11 bodies (three scenes)
D_LINKS, D_NSTEPS = 10, 36
D_PARAMS = [(0.7, [0.0, 1.0, 0.0]), (1.3, [0.0, 1.3, 0.1]), (0.5, [0.05, 0.9, -0.1])]


def chain(prefix="c_", params=C_PARAMS, links=C_LINKS, nsteps=C_NSTEPS):
    recs = [run_chain(m, f, links, nsteps) for m, f in params]
    out = {prefix + k: np.stack([r[k] for r in recs]) for k in recs[0]}
    out.update({prefix + "mass": np.array([m for m, _ in params]), prefix + "force": np.array([f for _, f in params]),
                prefix + "nsteps": np.int64(nsteps), prefix + "mult": np.float64(1500.0)})
    for i, r in enumerate(recs):
        print(prefix, params[i], "loss %.4f" % r["loss"], "grad mass", np.array2string(r["grad_mass"], precision=5), "grad force",
              np.array2string(r["grad_force"], precision=4), "steps with contact", np.nonzero(r["ncontacts"])[0].tolist(), "halved",
              int((np.diff(np.concatenate([[0.0], r["t"]])) < 0.99 / 30).sum()))
    return out


# initial positions of the "a_" scenes (the pendulum of "j_" with its bodies moved a little; gradients with respect to them)
A_POS0 = [([300.0, 300.0], [300.0, 372.0], [390.0, 300.0]), ([303.0, 298.0], [301.0, 371.0], [391.0, 302.0]),
          ([296.0, 303.0], [298.0, 374.0], [388.0, 297.0]), ([301.0, 305.0], [304.0, 370.0], [392.0, 301.0])]


def jointed(prefix, make, forces, pos0=None):
    recs = [run_joints(make, a, b, None if pos0 is None else pos0[i]) for i, (a, b) in enumerate(forces)]
    out = {prefix + k: np.stack([r[k] for r in recs]) for k in recs[0]}
    out.update({prefix + "force_first": np.array([a for a, _ in forces]), prefix + "force_ball": np.array([b for _, b in forces]),
                prefix + "nsteps": np.int64(J_NSTEPS)})
    for i, r in enumerate(recs):
        print(prefix, forces[i], "loss %.4f" % r["loss"], "grad first", np.array2string(r["grad_first"], precision=4), "grad ball",
              np.array2string(r["grad_ball"], precision=4), "steps with contact", np.nonzero(r["ncontacts"])[0].tolist(), "halved",
              int((np.diff(np.concatenate([[0.0], r["t"]])) < 0.99 / 30).sum()))
    return out


def main():
    ref_shim.load_reference()
    torch.set_default_dtype(torch.float64)
    jout = jointed("j_", make_world_pendulum, J_FORCES)
    jout.update(jointed("k_", make_world_dumbbell, K_FORCES))
    jout.update(jointed("a_", make_world_pendulum, J_FORCES[:len(A_POS0)], A_POS0))
    jout.update(chain())
    jout.update(chain("d_", D_PARAMS, D_LINKS, D_NSTEPS))
    hrecs = [run_hulls(a, b) for a, b in H_FORCES]
    hout = {"h_" + k: np.stack([r[k] for r in hrecs]) for k in hrecs[0]}
    hout.update(h_force_ball=np.array([a for a, _ in H_FORCES]), h_force_box=np.array([b for _, b in H_FORCES]), h_nsteps=np.int64(H_NSTEPS))
    for i, r in enumerate(hrecs):
        print("hulls", H_FORCES[i], "loss %.4f" % r["loss"], "grad ball", np.array2string(r["grad_ball"], precision=4), "grad box",
              np.array2string(r["grad_box"], precision=4), "max |box rot| %.1e" % r["box_rot_max"], "contacts", r["ncontacts"].tolist(), "halved",
              int((np.diff(np.concatenate([[0.0], r["t"]])) < 0.99 / 30).sum()))
    recs = [run(f) for f in FORCES]
    out = {k: np.stack([r[k] for r in recs]) for k in recs[0]}
    out.update(hout)
    out.update(jout)
    out.update(force0=np.array(FORCES), nsteps=np.int64(NSTEPS), t_push=np.float64(T_PUSH), mult=np.float64(MULT),
               dt=np.float64(1.0 / 30), no_contact=np.array([[0, 1], [0, 2]]), pushed_body=np.int64(1), loss_bodies=np.array([0, 2]))
    for i, r in enumerate(recs):
        print("force", FORCES[i], "loss %.4f" % r["loss"], "grad", np.array2string(r["grad"], precision=5),
              "steps with contact", int((r["ncontacts"] > 0).sum()), "halved", int((np.diff(np.concatenate([[0.0], r["t"]])) < 0.99 / 30).sum()))
    np.savez_compressed(os.path.join(OUT, "rollout_grad.npz"), **out)
    print("wrote", os.path.join(OUT, "rollout_grad.npz"))


if __name__ == "__main__":
    main()

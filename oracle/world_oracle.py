"""CPU oracle for one simulation step of a scene WITH contact detection.  TEST INFRASTRUCTURE ONLY.

Clean-room restatement (float64, one scene at a time) of `/root/reference/lcp_physics/physics`:

  world.py:83-122   World.step_dt           -> `step_dt`   (solve, move, find contacts, penetration check,
                                                           dt halving and reset; post-stabilisation, off by
                                                           default - utils.py:30 - behind `post_stab`)
  engines.py:80-116 PdipmEngine.post_stabilization -> `post_stabilization`
  engines.py:26-78  PdipmEngine.solve_dynamics -> `solve_dynamics` (both branches: the direct KKT solve when
                                                           there is no contact, :36-50, and the mixed LCP, :51-76)
  bodies.py:80-82   Body.move               -> inside `move_and_find`
  world.py:139-142  find_contacts           -> oracle.contacts_oracle.find_contacts

It is composed from the two oracles that are pinned on reference outputs by themselves
(`pdipm_oracle`, `contacts_oracle`) and is pinned as a whole on trajectories of the unmodified reference
`World` (`tests/golden/world_traj.npz`, made by `oracle/make_golden_world.py`) in
`tests/test_world_oracle.py`.

  constraints.py:13-217, world.py:156-170  the joints' J() / move() and World.Je -> `joint_jacobian`, `joints_move`

Joints: either a constant `Je`, or a `joints` dict (jtype, jb1, jb2, jr1, jrot1 - the encoding of
`lcp_joint_jacobian_f64`) whose Jacobian follows the pose: `Joint` (revolute), `FixedJoint`, X / Y / Rot / Total
constraints.  Forces: one array per call (the caller evaluates `force_func(t)`, forces.py:29-48).
"""
import numpy as np
import torch

from . import contacts_oracle as C
from . import pdipm_oracle as O


def bodies_at(shapes, p):
    """Contact-oracle bodies for pose p [nb,3] = (rot, x, y); shapes: ('circle', r) | ('rect', (w,h)) | ('hull', verts)."""
    out = []
    for (kind, a), q in zip(shapes, np.asarray(p, dtype=np.float64)):
        pos = np.array([q[1], q[2]])
        if kind == "circle":
            out.append(dict(kind="circle", pos=pos, rad=float(a)))
        elif kind == "rect":
            out.append(dict(kind="hull", pos=pos, verts=C.rect_verts(a, float(q[0]))))
        else:
            R = C.rotation_matrix(float(q[0]))                                   # bodies.py:211-214
            out.append(dict(kind="hull", pos=pos, verts=np.asarray(a, dtype=np.float64) @ R.T))
    return out


def solve_dynamics(Mdiag, v, f, dt, contacts, rest, fric, Je, max_iter=10):
    """engines.py:26-78 for one scene.  Mdiag, v, f: [nb,3]; contacts: reference-format list; Je: [e, 3nb] or None.
    Returns new_v [nb,3] (numpy float64)."""
    Mdiag, v, f = (np.asarray(t, dtype=np.float64) for t in (Mdiag, v, f))
    nb = v.shape[0]
    nz = 3 * nb
    e = 0 if Je is None else int(np.asarray(Je).shape[0])
    u = Mdiag.reshape(-1) * v.reshape(-1) + dt * f.reshape(-1)                   # :31-32
    if not contacts:                                                             # :36-50
        M = np.diag(Mdiag.reshape(-1))
        if e > 0:
            Jm = np.asarray(Je, dtype=np.float64)
            P = np.block([[M, -Jm.T], [Jm, np.zeros((e, e))]])
            x = np.linalg.solve(P, np.concatenate([u, np.zeros(e)]))
        else:
            x = u / Mdiag.reshape(-1)
        return x[:nz].reshape(nb, 3)
    t = lambda a, dt_=torch.float64: torch.as_tensor(np.asarray(a), dtype=dt_).unsqueeze(0)
    n = t(np.stack([c[0][0] for c in contacts]))
    p1 = t(np.stack([c[0][1] for c in contacts]))
    p2 = t(np.stack([c[0][2] for c in contacts]))
    i1 = t(np.array([c[1] for c in contacts]), torch.int64)
    i2 = t(np.array([c[2] for c in contacts]), torch.int64)
    Jet = t(Je) if e > 0 else torch.zeros(1, 0, nz, dtype=torch.float64)
    new_v, _, _ = O.solve_dynamics(t(Mdiag), t(v), t(f), dt, n, p1, p2, i1, i2, t(rest), t(fric), Jet,
                                   max_iter=max_iter)
    return new_v[0].numpy()


def post_stabilization(Mdiag, v, contacts, rest, Je, max_iter=10):
    """engines.py:80-116 for one scene: dp [nb,3].  (The reference needs at least one joint row here: `Je v` at :86.)"""
    Mdiag, v = (np.asarray(t, dtype=np.float64) for t in (Mdiag, v))
    nb = v.shape[0]
    nz = 3 * nb
    e = 0 if Je is None else int(np.asarray(Je).shape[0])
    if not contacts:                                                             # :92-103
        ge = np.asarray(Je, dtype=np.float64) @ v.reshape(-1) if e > 0 else np.zeros(0)
        M = np.diag(Mdiag.reshape(-1))
        if e > 0:
            Jm = np.asarray(Je, dtype=np.float64)
            P = np.block([[M, -Jm.T], [Jm, np.zeros((e, e))]])
        else:
            P = M
        x = np.linalg.solve(P, np.concatenate([np.zeros(nz), ge]))
        return -x[:nz].reshape(nb, 3)
    t = lambda a, dt_=torch.float64: torch.as_tensor(np.asarray(a), dtype=dt_).unsqueeze(0)
    n = t(np.stack([c[0][0] for c in contacts]))
    p1 = t(np.stack([c[0][1] for c in contacts]))
    p2 = t(np.stack([c[0][2] for c in contacts]))
    i1 = t(np.array([c[1] for c in contacts]), torch.int64)
    i2 = t(np.array([c[2] for c in contacts]), torch.int64)
    Jet = t(Je) if e > 0 else None
    dp, _, _ = O.post_stabilization(t(Mdiag), t(v), n, p1, p2, i1, i2, t(rest), Jet, max_iter=max_iter)
    return dp[0].numpy()


def move_and_find(shapes, p_start, v, dt, eps=0.1, tol=1e-6, strict=True, dt_floor=None, max_trials=64,
                  no_contact=()):
    """world.py:88-101: returns (p, contacts, dt_used, trials)."""
    p_start = np.asarray(p_start, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    dt_floor = dt / 4 if dt_floor is None else dt_floor
    trials = 0
    while True:
        p = p_start + v * dt                                                     # bodies.py:80-82
        contacts = C.find_contacts(bodies_at(shapes, p), eps=eps, no_contact=no_contact)
        trials += 1
        if all(c[0][3] <= tol for c in contacts):
            break
        if not strict and dt < dt_floor:
            break
        if trials >= max_trials:
            break
        dt = dt / 2
    return p, contacts, dt, trials


JOINT, FIXED, XCON, YCON, ROTCON, TOTAL = 1, 2, 3, 4, 5, 6
JOINT_ROWS = {JOINT: 2, FIXED: 3, XCON: 1, YCON: 1, ROTCON: 1, TOTAL: 3}


def joint_rows(jtype):
    return int(sum(JOINT_ROWS[int(t)] for t in jtype))


def joint_jacobian(joints, p):
    """World.Je (world.py:156-170) over the joints' J() (constraints.py): [e, 3 nb] for pose p [nb,3] = (rot, x, y).
    `Joint` (:13-36): pos1 = r1 (cos rot1, sin rot1) (polar_to_cart, utils.py:85-90), pos2 = body1.pos + pos1 - body2.pos,
    J1 = [[-pos1_y, 1, 0], [pos1_x, 0, 1]], J2 = [[pos2_y, -1, 0], [-pos2_x, 0, -1]];  `FixedJoint` (:56-79): pos1 = 0,
    pos2 = body1.pos - body2.pos, the same two rows plus [1, 0, 0] / [-1, 0, 0];  X / Y / Rot constraints (:95-172): one unit
    row;  TotalConstraint (:175-192): the 3 x 3 identity."""
    p = np.asarray(p, dtype=np.float64)
    nb = p.shape[0]
    Je = np.zeros((joint_rows(joints["jtype"]), 3 * nb))
    row = 0
    for k, t in enumerate(joints["jtype"]):
        t, b1, b2 = int(t), int(joints["jb1"][k]), int(joints["jb2"][k])
        if t in (JOINT, FIXED):
            pos1 = joints["jr1"][k] * np.array([np.cos(joints["jrot1"][k]), np.sin(joints["jrot1"][k])]) if t == JOINT else np.zeros(2)
            Je[row, 3 * b1:3 * b1 + 3] = [-pos1[1], 1, 0]
            Je[row + 1, 3 * b1:3 * b1 + 3] = [pos1[0], 0, 1]
            if b2 >= 0:
                pos2 = p[b1, 1:] + pos1 - p[b2, 1:]
                Je[row, 3 * b2:3 * b2 + 3] = [pos2[1], -1, 0]
                Je[row + 1, 3 * b2:3 * b2 + 3] = [-pos2[0], 0, -1]
            if t == FIXED:
                Je[row + 2, 3 * b1] = 1
                if b2 >= 0:
                    Je[row + 2, 3 * b2] = -1
        elif t == XCON:
            Je[row, 3 * b1 + 1] = 1
        elif t == YCON:
            Je[row, 3 * b1 + 2] = 1
        elif t == ROTCON:
            Je[row, 3 * b1] = 1
        elif t == TOTAL:
            Je[row:row + 3, 3 * b1:3 * b1 + 3] = np.eye(3)
        row += JOINT_ROWS[t]
    return Je


def joints_move(joints, v, dt):
    """Joint.move (constraints.py:39-43): rot1 += body1.v[0] dt, from the rot1 the step started with (world.py:84,102-107 reset
    it before every retry).  Returns a new joints dict."""
    out = dict(joints)
    rot = np.array(joints["jrot1"], dtype=np.float64).copy()
    for k, t in enumerate(joints["jtype"]):
        if int(t) == JOINT:
            rot[k] = rot[k] + np.asarray(v, dtype=np.float64)[int(joints["jb1"][k]), 0] * dt
    out["jrot1"] = rot
    return out


def step_dt(shapes, p, v, contacts, Mdiag, f, rest, fric, Je, dt, eps=0.1, tol=1e-6, strict=True, max_iter=10,
            no_contact=(), post_stab=False, joints=None):
    """world.py:83-122.  Returns (p_new, v_new, contacts_new, dt_used, trials) - and the moved joints as a sixth item when
    `joints` (pose-dependent Jacobian) is given instead of a constant `Je`."""
    if joints is not None:
        Je = joint_jacobian(joints, p)
    new_v = solve_dynamics(Mdiag, v, f, dt, contacts, rest, fric, Je, max_iter=max_iter)
    p_new, cs, dt_used, trials = move_and_find(shapes, p, new_v, dt, eps=eps, tol=tol, strict=strict,
                                               dt_floor=dt / 4, no_contact=no_contact)
    if joints is not None:
        joints = joints_move(joints, new_v, dt_used)
    if post_stab:                                                                # :109-121
        if joints is not None:
            Je = joint_jacobian(joints, p_new)
        dp = post_stabilization(Mdiag, new_v, cs, rest, Je) / 2
        p_new = p_new + dp * dt_used
        if joints is not None:
            joints = joints_move(joints, dp, dt_used)
        cs = C.find_contacts(bodies_at(shapes, p_new), eps=eps, no_contact=no_contact)
    if joints is not None:
        return p_new, new_v, cs, dt_used, trials, joints
    return p_new, new_v, cs, dt_used, trials

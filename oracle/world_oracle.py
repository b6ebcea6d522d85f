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

Restrictions (same as the device path it checks): joints must have a constant Jacobian
(TotalConstraint / XConstraint / YConstraint / RotConstraint rows - `constraints.py`), forces are constant.
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


def step_dt(shapes, p, v, contacts, Mdiag, f, rest, fric, Je, dt, eps=0.1, tol=1e-6, strict=True, max_iter=10,
            no_contact=(), post_stab=False):
    """world.py:83-122.  Returns (p_new, v_new, contacts_new, dt_used, trials)."""
    new_v = solve_dynamics(Mdiag, v, f, dt, contacts, rest, fric, Je, max_iter=max_iter)
    p_new, cs, dt_used, trials = move_and_find(shapes, p, new_v, dt, eps=eps, tol=tol, strict=strict,
                                               dt_floor=dt / 4, no_contact=no_contact)
    if post_stab:                                                                # :109-121
        dp = post_stabilization(Mdiag, new_v, cs, rest, Je) / 2
        p_new = p_new + dp * dt_used
        cs = C.find_contacts(bodies_at(shapes, p_new), eps=eps, no_contact=no_contact)
    return p_new, new_v, cs, dt_used, trials

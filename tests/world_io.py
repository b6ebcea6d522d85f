"""Shared loader of tests/golden/world_traj.npz (reference `World` trajectories, oracle/make_golden_world.py)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_world_traj():
    d = np.load(os.path.join(GOLD, "world_traj.npz"))
    out = {}
    for name in d["names"].tolist():
        out[name] = {k.split("__", 1)[1]: d[k] for k in d.files if k.startswith(name + "__")}
    return out


def shapes_of(rec):
    return [("circle", float(s[0])) if int(k) == 0 else ("rect", (float(s[0]), float(s[1])))
            for k, s in zip(rec["kind"], rec["size"])]


class _RecordedBody:
    def __init__(self, rest, fric):
        self.restitution, self.fric_coeff = rest, fric


class RecordedWorld:
    """Replays what the reference's REAL `World` answered when its engine asked (oracle/make_golden.py records, inside
    `PdipmEngine.solve_dynamics`, the outputs of `world.M()`, `get_v()`, `apply_forces(t)`, `Je()`, `world.contacts`
    and `bodies[i].restitution / .fric_coeff` at that step; `/root/reference` cannot travel to the GPU box).  Nothing is
    reconstructed by hand: every accessor returns the recorded tensor.  `leaf=True` makes the tensors autograd leaves
    (what `requires_grad` parameters of the world's bodies / forces would be)."""

    vec_len = 3
    static_inverse = True
    post_stab = False

    def __init__(self, st, with_contacts=True, leaf=False):
        import torch
        mk = (lambda t: t.clone().double().requires_grad_(True)) if leaf else (lambda t: t.double())
        self.t, self.dt = float(st["t"]) if "t" in st else 0.0, st["dt"]
        self._v, self._f, self._Md = mk(st["v"]), mk(st["f"]), mk(st["Mdiag"])
        self._rest, self._fric = mk(st["rest"]), mk(st["fric"])
        self._Je = mk(st["Je"])                          # (a leaf too: the engine's backward returns d(loss)/dJe, lcp.py:57)
        nb = st["v"].shape[0]
        self.bodies = [_RecordedBody(self._rest[i], self._fric[i]) for i in range(nb)]
        self._cn, self._cp1, self._cp2 = mk(st["c_n"]), mk(st["c_p1"]), mk(st["c_p2"])
        self.contacts = []
        if with_contacts:
            self.contacts = [((self._cn[i], self._cp1[i], self._cp2[i], st["c_pen"][i]), int(st["c_i1"][i]), int(st["c_i2"][i]))
                             for i in range(st["c_n"].shape[0])]
        self._torch = torch

    def M(self):
        return self._torch.diag(self._Md.reshape(-1))

    def get_v(self):
        return self._v.reshape(-1)

    def apply_forces(self, t):
        return self._f.reshape(-1)

    def Je(self):
        return self._Je

    def leaves(self):
        return {"Mdiag": self._Md, "v": self._v, "f": self._f, "rest": self._rest, "fric": self._fric,
                "c_n": self._cn, "c_p1": self._cp1, "c_p2": self._cp2}

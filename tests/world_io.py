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

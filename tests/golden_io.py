"""Loader for the committed fixtures in tests/golden/ (made by oracle/make_golden.py)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES = ["ball_floor", "stack2", "stack4", "slide", "fric"]


def _t(a):
    return torch.from_numpy(np.asarray(a))


def load_steps(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    steps = []
    for k in range(int(d["n_steps"])):
        pfx = "s%d_" % k
        st = {key[len(pfx):]: _t(d[key]) for key in d.files if key.startswith(pfx)}
        st["dt"] = float(d[pfx + "dt"])
        steps.append(st)
    return steps


def all_steps():
    for name in SCENES:
        for k, st in enumerate(load_steps(name)):
            yield "%s[%d]" % (name, k), st


def lcp_inputs(st, dtype=torch.float64):
    out = []
    for nm in "QpGhAbF":
        a = st["in_" + nm]
        out.append(a.to(dtype) if a.numel() else None)
    return out


def ref_grads(st):
    return {nm: st.get("grad_" + nm) for nm in "QpGhAbF"}

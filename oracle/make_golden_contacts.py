"""Generate tests/golden/contacts_pairs.npz and contacts_scenes.npz by running the UNMODIFIED reference
contact handler (`lcp_physics/physics/contacts.py:50-205`, through oracle/ref_shim.py) on seeded random
configurations.  TEST INFRASTRUCTURE ONLY; needs /root/reference.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_contacts.py
"""
import math
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


class _FakeWorld:
    def __init__(self, bodies, eps=0.1):
        self.bodies, self.eps, self.contacts = bodies, eps, []
        for i, b in enumerate(bodies):
            b.geom.body = i


def _make(kind, pos3, size):
    from lcp_physics.physics.bodies import Circle, Rect
    if kind == 0:
        return Circle([pos3[1], pos3[2]], size[0])
    return Rect([pos3[0], pos3[1], pos3[2]], [size[0], size[1]])


def _run_pairs(world):
    from lcp_physics.physics.contacts import DiffContactHandler
    h = DiffContactHandler()
    n = len(world.bodies)
    for i in range(n):
        for j in range(i + 1, n):
            h([world], world.bodies[i].geom, world.bodies[j].geom)
    return world.contacts


def gen_pairs(n, rng):
    rec = dict(kind=[], pos=[], size=[], count=[], normal=[], p1=[], p2=[], pen=[])
    while len(rec["count"]) < n:
        k1, k2 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        s1 = [float(rng.uniform(10, 40)), float(rng.uniform(10, 40))]
        s2 = [float(rng.uniform(10, 40)), float(rng.uniform(10, 40))]
        if rng.random() < 0.3:            # axis-aligned boxes resting on each other (the stack case)
            r1 = r2 = 0.0
        else:
            r1, r2 = float(rng.uniform(-math.pi, math.pi)), float(rng.uniform(-math.pi, math.pi))
        ext1 = s1[0] if k1 == 0 else 0.5 * math.hypot(*s1)
        ext2 = s2[0] if k2 == 0 else 0.5 * math.hypot(*s2)
        ang = float(rng.uniform(0, 2 * math.pi))
        # centre distance from deep overlap to clearly separated
        mode = rng.random()
        if k1 == 1 and k2 == 1 and r1 == 0.0 and mode < 0.6:
            dx = float(rng.uniform(-0.9, 0.9)) * 0.5 * (s1[0] + s2[0])
            dy = 0.5 * (s1[1] + s2[1]) + float(rng.uniform(-0.3, 0.25))
            off = np.array([dx, dy if rng.random() < 0.5 else -dy])
        else:
            d = float(rng.uniform(0.15, 1.15)) * (ext1 + ext2)
            off = d * np.array([math.cos(ang), math.sin(ang)])
        p1 = [r1, 300.0, 300.0]
        p2 = [r2, 300.0 + float(off[0]), 300.0 + float(off[1])]
        random.seed(len(rec["count"]))
        b = [_make(k1, p1, s1), _make(k2, p2, s2)]
        try:
            cs = _run_pairs(_FakeWorld(b))
        except Exception:                  # (reference raises on some degenerate GJK configurations)
            continue
        if len(cs) == 0 and rng.random() < 0.6:
            continue                       # keep the set contact-rich
        rec["kind"].append([k1, k2]); rec["pos"].append([p1, p2]); rec["size"].append([s1, s2])
        rec["count"].append(len(cs))
        pad = lambda key, k: [c[0][k].detach().numpy().reshape(-1) for c in cs] + [np.zeros(2 if k < 3 else 1)] * (2 - len(cs))
        rec["normal"].append(np.stack(pad("n", 0))); rec["p1"].append(np.stack(pad("p1", 1)))
        rec["p2"].append(np.stack(pad("p2", 2)))
        rec["pen"].append(np.array([float(c[0][3]) for c in cs] + [0.0] * (2 - len(cs))))
    return {k: np.asarray(v) for k, v in rec.items()}


def gen_scenes(n, rng):
    """Small multi-body scenes: floor + boxes / balls dropped near each other (contact lists with order)."""
    out = []
    for s in range(n):
        nb = int(rng.integers(3, 6))
        kinds, poss, sizes = [1], [[0.0, 300.0, 400.0]], [[500.0, 10.0]]
        y = 395.0
        for i in range(nb - 1):
            k = int(rng.integers(0, 2))
            sz = [float(rng.uniform(15, 30)), float(rng.uniform(15, 30))]
            hh = sz[0] if k == 0 else sz[1] / 2
            y -= hh
            x = 300.0 + float(rng.uniform(-12, 12))
            rot = 0.0 if rng.random() < 0.7 else float(rng.uniform(-0.05, 0.05))
            poss.append([rot, x, y + float(rng.uniform(-0.05, 0.08))])
            kinds.append(k); sizes.append(sz)
            y -= hh
        random.seed(1000 + s)
        bodies = [_make(k, p, z) for k, p, z in zip(kinds, poss, sizes)]
        try:
            cs = _run_pairs(_FakeWorld(bodies))
        except Exception:
            continue
        out.append(dict(kind=np.array(kinds), pos=np.array(poss), size=np.array(sizes),
                        i1=np.array([c[1] for c in cs], dtype=np.int32), i2=np.array([c[2] for c in cs], dtype=np.int32),
                        normal=np.array([c[0][0].detach().numpy() for c in cs]).reshape(-1, 2),
                        p1=np.array([c[0][1].detach().numpy() for c in cs]).reshape(-1, 2),
                        p2=np.array([c[0][2].detach().numpy() for c in cs]).reshape(-1, 2),
                        pen=np.array([float(c[0][3]) for c in cs])))
    return out


def main():
    ref_shim.load_reference()
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(2024)
    pairs = gen_pairs(600, rng)
    np.savez_compressed(os.path.join(OUT, "contacts_pairs.npz"), **pairs)
    print("pairs:", {k: v.shape for k, v in pairs.items()}, "count histogram", np.bincount(pairs["count"]))
    scenes = gen_scenes(60, rng)
    flat = {}
    for i, sc in enumerate(scenes):
        for k, v in sc.items():
            flat["s%d_%s" % (i, k)] = v
    flat["n"] = np.int64(len(scenes))
    np.savez_compressed(os.path.join(OUT, "contacts_scenes.npz"), **flat)
    print("scenes:", len(scenes), "contacts per scene", [len(s["pen"]) for s in scenes][:20])


if __name__ == "__main__":
    main()

"""CPU: pin the contact-generation oracle (oracle/contacts_oracle.py) against the reference's
DiffContactHandler outputs recorded in tests/golden/contacts_*.npz (oracle/make_golden_contacts.py)."""
import os

import numpy as np
import pytest

from oracle import contacts_oracle as C

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def body(kind, pos3, size):
    if int(kind) == 0:
        return dict(kind="circle", pos=np.array(pos3[1:], dtype=np.float64), rad=float(size[0]))
    return dict(kind="hull", pos=np.array(pos3[1:], dtype=np.float64), verts=C.rect_verts(size, float(pos3[0])))


def test_pairs_match_reference_fixture():
    d = np.load(os.path.join(GOLD, "contacts_pairs.npz"))
    n = len(d["count"])
    kinds_seen = set()
    for i in range(n):
        b1 = body(d["kind"][i, 0], d["pos"][i, 0], d["size"][i, 0])
        b2 = body(d["kind"][i, 1], d["pos"][i, 1], d["size"][i, 1])
        pts = C.collide_pair(b1, b2, eps=0.1)
        assert len(pts) == int(d["count"][i]), (i, len(pts), int(d["count"][i]))
        kinds_seen.add((int(d["kind"][i, 0]), int(d["kind"][i, 1]), len(pts)))
        for k, (nrm, p1, p2, pen) in enumerate(pts):
            assert np.allclose(nrm, d["normal"][i, k], atol=1e-9), (i, k, "normal")
            assert np.allclose(p1, d["p1"][i, k], atol=1e-8), (i, k, "p1")
            assert np.allclose(p2, d["p2"][i, k], atol=1e-8), (i, k, "p2")
            assert abs(pen - d["pen"][i, k]) < 1e-8, (i, k, "pen")
    # the fixture covers every shape pair with and without contact, and two-point manifolds
    for kk in ((0, 0), (0, 1), (1, 0), (1, 1)):
        assert (kk[0], kk[1], 1) in kinds_seen
    assert (1, 1, 2) in kinds_seen and any(k[2] == 0 for k in kinds_seen)


def test_scene_contact_lists_match_reference_fixture():
    d = np.load(os.path.join(GOLD, "contacts_scenes.npz"))
    total = 0
    for s in range(int(d["n"])):
        g = lambda k: d["s%d_%s" % (s, k)]
        bodies = [body(k, p, z) for k, p, z in zip(g("kind"), g("pos"), g("size"))]
        cs = C.find_contacts(bodies, eps=0.1)
        assert [c[1] for c in cs] == g("i1").tolist() and [c[2] for c in cs] == g("i2").tolist(), s
        for k, c in enumerate(cs):
            assert np.allclose(c[0][0], g("normal")[k], atol=1e-9)
            assert np.allclose(c[0][1], g("p1")[k], atol=1e-8) and np.allclose(c[0][2], g("p2")[k], atol=1e-8)
            assert abs(c[0][3] - g("pen")[k]) < 1e-8
        total += len(cs)
    assert total > 100

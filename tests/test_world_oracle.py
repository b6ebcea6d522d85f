"""CPU: pin oracle/world_oracle.py (step_dt = engine solve + move + contact detection + dt halving) on
trajectories of the unmodified reference World (tests/golden/world_traj.npz)."""
import numpy as np
import pytest

from oracle import contacts_oracle as C
from oracle import world_oracle as W
from tests.world_io import load_world_traj, shapes_of

TRAJ = load_world_traj()


@pytest.mark.parametrize("name", sorted(TRAJ))
def test_world_oracle_follows_reference_trajectory(name):
    rec = TRAJ[name]
    shapes = shapes_of(rec)
    dt, strict = float(rec["dt"]), bool(rec["strict"])
    p, v, t = rec["p"][0].copy(), rec["v"][0].copy(), 0.0
    contacts = C.find_contacts(W.bodies_at(shapes, p), eps=float(rec["eps"]))          # World.__init__ (world.py:65-66)
    assert len(contacts) == int(rec["ncontacts"][0])
    halved = 0
    for k in range(1, len(rec["t"])):
        p, v, contacts, dt_used, trials = W.step_dt(shapes, p, v, contacts, rec["Mdiag"], rec["f"], rec["rest"],
                                                    rec["fric"], rec["Je"], dt, eps=float(rec["eps"]),
                                                    tol=float(rec["tol"]), strict=strict)
        t += dt_used
        halved += trials > 1
        assert abs(t - rec["t"][k]) < 1e-12, (name, k, "t")
        assert len(contacts) == int(rec["ncontacts"][k]), (name, k, "contact count")
        assert np.allclose(v, rec["v"][k], atol=1e-6, rtol=1e-7), (name, k, "v", np.abs(v - rec["v"][k]).max())
        assert np.allclose(p, rec["p"][k], atol=1e-6, rtol=1e-9), (name, k, "p", np.abs(p - rec["p"][k]).max())
    assert halved > 0          # every fixture exercises the dt-halving loop

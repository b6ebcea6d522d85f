"""Throughput of a batched world of chains: the reference's chain scene (five links hinged by revolute `Joint`s, 10 equality rows,
hit by a projectile - tests/golden/world_traj.npz "chain", recorded from the unmodified reference) replicated B times,
`ContactWorld.step()` with the joint Jacobian rebuilt every step.  5 .. 16 equality rows run on lcp_primal.hip's 16-row
instantiation; `generic` forces the round-1 route (the generic kernel) for comparison.

    python tools/experiments/chain_world.py [--batch 4096] [--steps 30] [generic]
"""
import json, os, sys, time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from lcp_physics_amd import _lib
    from lcp_physics_amd.physics.batched_world import ContactWorld
    from lcp_physics_amd.physics.contacts import GeometryBatch
    from lcp_physics_amd.physics.joints import JointSet
    from tests.world_io import load_world_traj, shapes_of
    args = sys.argv[1:]
    B = int(args[args.index("--batch") + 1]) if "--batch" in args else 4096
    steps = int(args[args.index("--steps") + 1]) if "--steps" in args else 30
    path = "generic" if "generic" in args else "auto"
    dev = "cuda"
    rec = load_world_traj()["chain"]
    shapes = shapes_of(rec)
    nb = len(shapes)
    geom = GeometryBatch.from_shapes(shapes, B)
    if rec["no_contact"].size:
        nocon = torch.zeros(B, nb, nb, dtype=torch.uint8)
        for i, j in rec["no_contact"].tolist():
            nocon[:, i, j] = 1
        geom.no_contact = nocon
    geom = geom.to(dev)
    rep = lambda a, dt_: torch.tensor(np.broadcast_to(a, (B,) + a.shape).copy(), dtype=dt_, device=dev)
    f_t = torch.tensor(rec["f_t"], dtype=torch.float32, device=dev)
    k_ = {"k": 0}
    res = {}
    for label in (path,):
        joints = JointSet.from_arrays(rec["jtype"], rec["jb1"], rec["jb2"], rec["jr1"], rec["jrot1"], B).to(dev)
        world = ContactWorld(geom, rep(rec["p"][0], torch.float64), rep(rec["v"][0], torch.float32), rep(rec["Mdiag"], torch.float32),
                             rep(rec["f"], torch.float32), rep(rec["rest"], torch.float32), rep(rec["fric"], torch.float32),
                             dt=float(rec["dt"]), eps=float(rec["eps"]), tol=float(rec["tol"]), strict_no_penetration=bool(rec["strict"]),
                             maxc=8, joints=joints, force_fn=lambda t: f_t[min(k_["k"], f_t.shape[0] - 1)].unsqueeze(0).expand(B, -1, -1))
        _lib.set_path(label)
        try:
            nrec = len(rec["t"]) - 1
            for k in range(3):
                k_["k"] = k
                world.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(3, 3 + steps):
                k_["k"] = k
                world.step()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        finally:
            _lib.set_path("auto")
        ok = None
        if 3 + steps <= nrec:
            ok = float(np.abs(world.p[0].cpu().numpy() - rec["p"][3 + steps]).max())
        res = {"metric": "sim steps/s, chain world (5 bodies, 10 equality rows, contacts <= 8)", "path": label, "value": B * steps / wall,
               "ms_per_step": wall / steps * 1e3, "batch": B, "steps": steps, "e": int(world.e), "worst |p - reference p| of scene 0": ok}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

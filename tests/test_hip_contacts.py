"""GPU parity tests of the contact-generation / position-update kernel (`lcp_move_find_contacts_f64`), the
variable-contact-count engine solve (`lcp_solve_dynamics_f32`) and the batched `ContactWorld`, through the C ABI,
against the reference fixtures (tests/golden/contacts_*.npz, world_traj.npz) and the CPU oracles.

Tolerances: contact index lists, counts, trial counts: exact.  Penetration (fp64 end to end): 1e-9.
Contact frame (normal, arms): the kernel computes them in fp64 and rounds to fp32 for the LCP kernels, so they
are compared at fp32 resolution of their magnitude (|arm| <= ~500 -> 1e-4 absolute; normals 1e-6).
"""
import os

import numpy as np
import pytest
import torch

from tests import parity

from oracle import contacts_oracle as C
from oracle import world_oracle as W
from tests.world_io import load_world_traj, shapes_of

pytestmark = pytest.mark.gpu
DEV = "cuda"
ATOL_ARM, ATOL_N, ATOL_PEN = 1e-4, 1e-6, 1e-9


def _geom(shape_lists):
    """GeometryBatch for a list (one per scene) of shape lists of equal length."""
    from lcp_physics_amd.physics.contacts import GeometryBatch
    gs = [GeometryBatch.from_shapes(sh, 1) for sh in shape_lists]
    cat = lambda k: torch.cat([getattr(g, k) for g in gs])
    return GeometryBatch(cat("kind"), cat("radius"), cat("verts_local"), cat("nverts"), None).to(DEV)


def _shape(kind, size):
    return ("circle", float(size[0])) if int(kind) == 0 else ("rect", (float(size[0]), float(size[1])))


def _compare_lists(cb, k, ref, what):
    """Contact list of scene k in the device buffers against a reference-format list."""
    n = int(cb.count[k])
    assert n == len(ref), (what, "count", n, len(ref))
    if n == 0:
        return
    i1, i2 = cb.c_i1[k, :n].cpu().tolist(), cb.c_i2[k, :n].cpu().tolist()
    assert i1 == [c[1] for c in ref] and i2 == [c[2] for c in ref], (what, "body indices")
    g = lambda t: t[k, :n].double().cpu().numpy()
    rn, r1, r2 = (np.stack([c[0][q] for c in ref]) for q in range(3))
    rp = np.array([c[0][3] for c in ref])
    assert np.abs(g(cb.c_n) - rn).max() <= ATOL_N, (what, "normal", np.abs(g(cb.c_n) - rn).max())
    assert np.abs(g(cb.c_p1) - r1).max() <= ATOL_ARM, (what, "p1", np.abs(g(cb.c_p1) - r1).max())
    assert np.abs(g(cb.c_p2) - r2).max() <= ATOL_ARM, (what, "p2", np.abs(g(cb.c_p2) - r2).max())
    assert np.abs(g(cb.c_pen) - rp).max() <= ATOL_PEN, (what, "penetration", np.abs(g(cb.c_pen) - rp).max())


def test_pairs_match_reference_fixture():
    """600 reference DiffContactHandler outputs (contacts.py:57-205), one pair per scene."""
    from lcp_physics_amd.physics.contacts import find_contacts
    from tests.test_contacts_oracle import GOLD
    import os
    d = np.load(os.path.join(GOLD, "contacts_pairs.npz"))
    n = len(d["count"])
    geom = _geom([[_shape(d["kind"][i, 0], d["size"][i, 0]), _shape(d["kind"][i, 1], d["size"][i, 1])] for i in range(n)])
    p = torch.tensor(d["pos"], dtype=torch.float64, device=DEV)
    cb = find_contacts(geom, p, maxc=4)
    torch.cuda.synchronize()
    assert cb.count.cpu().tolist() == d["count"].tolist()
    for i in range(n):
        ref = [((d["normal"][i, k], d["p1"][i, k], d["p2"][i, k], d["pen"][i, k]), 0, 1) for k in range(int(d["count"][i]))]
        _compare_lists(cb, i, ref, "pair %d" % i)
    # padded slots are zeroed
    pad = torch.arange(4, device=DEV).unsqueeze(0) >= cb.count.unsqueeze(1)
    assert float(cb.c_n[pad].abs().max()) == 0.0 and int(cb.c_i2[pad].abs().max()) == 0


def test_scene_lists_match_reference_fixture():
    """60 multi-body scenes: the order of the contact list is the reference's (pair order i < j)."""
    from lcp_physics_amd.physics.contacts import find_contacts
    from tests.test_contacts_oracle import GOLD
    import os
    d = np.load(os.path.join(GOLD, "contacts_scenes.npz"))
    by_nb = {}
    for s in range(int(d["n"])):
        by_nb.setdefault(len(d["s%d_kind" % s]), []).append(s)
    for nb, ids in by_nb.items():
        g = lambda s, k: d["s%d_%s" % (s, k)]
        geom = _geom([[_shape(k, z) for k, z in zip(g(s, "kind"), g(s, "size"))] for s in ids])
        p = torch.tensor(np.stack([g(s, "pos") for s in ids]), dtype=torch.float64, device=DEV)
        cb = find_contacts(geom, p, maxc=16)
        torch.cuda.synchronize()
        for k, s in enumerate(ids):
            ref = [((g(s, "normal")[q], g(s, "p1")[q], g(s, "p2")[q], g(s, "pen")[q]), int(g(s, "i1")[q]), int(g(s, "i2")[q]))
                   for q in range(len(g(s, "pen")))]
            _compare_lists(cb, k, ref, "scene %d" % s)


def _random_scene(rng, nb, hulls=True, rotate=True):
    """Floor + bodies dropped near each other; returns (shapes, pose[nb,3])."""
    shapes, pose = [("rect", (500.0, 10.0))], [[0.0, 300.0, 400.0]]
    y = 395.0
    for _ in range(nb - 1):
        r = rng.random()
        sz = rng.uniform(15, 30, size=2)
        if r < 0.35:
            shapes.append(("circle", float(sz[0]))); hh = sz[0]
        elif r < 0.8 or not hulls:
            shapes.append(("rect", (float(sz[0]), float(sz[1])))); hh = sz[1] / 2
        else:                                                   # convex polygon with 3..6 vertices (a Hull, bodies.py:135)
            nv = int(rng.integers(3, 7))
            # counter-clockwise like the reference's Rect (bodies.py:261-264): left_orthogonal(edge) points outwards
            ang = (np.arange(nv) + rng.uniform(-0.3, 0.3, nv)) * (2 * np.pi / nv) + rng.uniform(0, 2 * np.pi)
            rad = float(sz[0])
            shapes.append(("hull", np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=1))); hh = rad * 0.8
        y -= hh
        rot = 0.0 if (rng.random() < 0.5 or not rotate) else float(rng.uniform(-0.4, 0.4))
        pose.append([rot, 300.0 + float(rng.uniform(-15, 15)), y + float(rng.uniform(-0.3, 0.1))])
        y -= hh
    return shapes, np.array(pose)


def test_random_scenes_match_oracle():
    """Circles, rects and general convex hulls (3-6 vertices), 3-6 bodies, against oracle/contacts_oracle.py."""
    from lcp_physics_amd.physics.contacts import find_contacts
    rng = np.random.default_rng(7)
    total = 0
    for nb in (3, 4, 6, 7, 12):                  # 6 bodies: four scenes per wave; 7: one wave per scene; 12: 66 pairs > 64 lanes
        scenes = [_random_scene(rng, nb) for _ in range(97 if nb < 12 else 24)]
        geom = _geom([s[0] for s in scenes])
        p = torch.tensor(np.stack([s[1] for s in scenes]), dtype=torch.float64, device=DEV)
        cb = find_contacts(geom, p, maxc=48)
        torch.cuda.synchronize()
        for k, (shapes, pose) in enumerate(scenes):
            try:
                ref = C.find_contacts(W.bodies_at(shapes, pose), eps=0.1)
            except ValueError:          # get_closest raises on a degenerate simplex (contacts.py:330) - no reference answer
                continue
            _compare_lists(cb, k, ref, "nb%d scene %d" % (nb, k))
            total += len(ref)
    assert total > 400


def test_no_contact_mask_and_capacity_report():
    from lcp_physics_amd.physics.contacts import find_contacts
    rng = np.random.default_rng(3)
    shapes, pose = _random_scene(rng, 5, hulls=False)
    geom = _geom([shapes, shapes])
    ref0 = C.find_contacts(W.bodies_at(shapes, pose), eps=0.1)
    a, b = ref0[0][1], ref0[0][2]
    mask = torch.zeros(2, 5, 5, dtype=torch.uint8)
    mask[1, a, b] = 1                                            # scene 1 ignores one touching pair (bodies.py:117-118)
    geom.no_contact = mask.to(DEV)
    p = torch.tensor(np.stack([pose, pose]), dtype=torch.float64, device=DEV)
    cb = find_contacts(geom, p, maxc=16)
    ref1 = C.find_contacts(W.bodies_at(shapes, pose), eps=0.1, no_contact=[(a, b)])
    assert len(ref1) < len(ref0)
    _compare_lists(cb, 0, ref0, "unmasked")
    _compare_lists(cb, 1, ref1, "masked")
    # a list longer than the capacity is reported through `count` (and truncated, not overrun)
    small = find_contacts(geom, p, maxc=2)
    assert int(small.count[0]) == len(ref0) > 2
    assert small.c_n.shape[1] == 2


def test_move_and_halve_matches_oracle():
    """world.py:88-101: accepted dt, number of trials, pose and contact list per scene."""
    from lcp_physics_amd.physics.contacts import move_and_find_contacts
    rng = np.random.default_rng(11)
    nb, B = 4, 128
    scenes = [_random_scene(rng, nb, hulls=False, rotate=False) for _ in range(B)]
    # lift the bodies clear of each other and throw them down: most scenes penetrate at the full dt
    p0 = np.stack([s[1] for s in scenes])
    p0[:, 1:, 2] -= rng.uniform(0.5, 3.0, size=(B, nb - 1)).cumsum(axis=1)
    v = np.zeros((B, nb, 3))
    v[:, 1:, 2] = rng.uniform(20, 120, size=(B, nb - 1))
    v[:, 1:, 1] = rng.uniform(-20, 20, size=(B, nb - 1))
    v[:, 1:, 0] = rng.uniform(-0.5, 0.5, size=(B, nb - 1))
    v32 = torch.tensor(v, dtype=torch.float32)
    geom = _geom([s[0] for s in scenes])
    dt = 1.0 / 30
    for strict in (True, False):
        t = torch.zeros(B, dtype=torch.float64, device=DEV)
        cb = move_and_find_contacts(geom, torch.tensor(p0, dtype=torch.float64, device=DEV), v32.to(DEV), dt,
                                    maxc=16, strict=strict, t=t)
        torch.cuda.synchronize()
        halved = 0
        for k in range(B):
            p_ref, ref, dt_ref, trials = W.move_and_find(scenes[k][0], p0[k], v32[k].double().numpy(), dt, strict=strict)
            assert int(cb.trials[k]) == trials, (strict, k, int(cb.trials[k]), trials)
            assert float(cb.dt_used[k]) == dt_ref and abs(float(t[k]) - dt_ref) < 1e-15
            assert np.abs(cb.p_out[k].cpu().numpy() - p_ref).max() < 1e-10
            _compare_lists(cb, k, ref, "strict=%s scene %d" % (strict, k))
            halved += trials > 1
        assert halved > B // 4


def _variable_count_batch(rng, B, nb, maxc):
    """Random scenes with their oracle contact lists (0 .. maxc contacts)."""
    scenes, lists = [], []
    while len(scenes) < B:
        shapes, pose = _random_scene(rng, nb, hulls=False)
        if rng.random() < 0.2:
            pose[1:, 2] -= 40.0                                  # lifted clear of the floor: fewer / no contacts
        if rng.random() < 0.1:
            pose[1:, 2] -= 400.0 * np.arange(1, nb)              # everything apart: no contact at all
        try:
            cs = C.find_contacts(W.bodies_at(shapes, pose), eps=0.1)
        except ValueError:
            continue
        if len(cs) <= maxc:
            scenes.append((shapes, pose)); lists.append(cs)
    return scenes, lists


@pytest.fixture(params=["auto", "generic"])
def solver_path(request):
    """The four-scenes-per-wave kernel (auto, for these sizes) and the workgroup-per-scene kernels (any size)."""
    from lcp_physics_amd import _lib
    _lib.set_path(request.param)
    yield request.param
    _lib.set_path("auto")


@pytest.mark.parametrize("with_joint", [True, False])
def test_solve_dynamics_variable_counts_match_oracle(with_joint, solver_path):
    """engines.py:26-78 with per-scene contact counts (incl. the no-contact branch :36-50): new_v within 1e-4
    (scaled by the velocity scale of the scene) of the fp64 oracle."""
    from lcp_physics_amd.physics.batched_world import solve_dynamics
    from lcp_physics_amd.physics.contacts import find_contacts
    rng = np.random.default_rng(5)
    B, nb, maxc = (192, 4, 12) if solver_path == "auto" else (64, 4, 12)
    scenes, lists = _variable_count_batch(rng, B, nb, maxc)
    counts = [len(c) for c in lists]
    assert min(counts) == 0 and max(counts) >= 5 and len(set(counts)) >= 4
    geom = _geom([s[0] for s in scenes])
    p = torch.tensor(np.stack([s[1] for s in scenes]), dtype=torch.float64, device=DEV)
    cb = find_contacts(geom, p, maxc=maxc)
    mass = rng.uniform(0.5, 2.0, size=(B, nb))
    Mdiag = np.stack([mass * rng.uniform(50, 200, size=(B, nb)), mass, mass], axis=-1)
    v = rng.normal(0, 5.0, size=(B, nb, 3)); v[:, :, 0] *= 0.05
    f = np.zeros((B, nb, 3)); f[:, :, 2] = mass * 100.0
    rest, fric = rng.uniform(0.1, 0.6, size=(B, nb)), rng.uniform(0.2, 0.8, size=(B, nb))
    if with_joint:
        Je = np.zeros((B, 3, 3 * nb)); Je[:, :, :3] = np.eye(3)       # TotalConstraint on the floor (constraints.py:176-192)
        v[:, 0] = 0
    else:
        Je = None
        Mdiag[:, 0] *= 1e4; f[:, 0] = 0; v[:, 0] = 0                  # a heavy free floor
    g32 = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV).contiguous()
    Mg, vg, fg, rg, cg = g32(Mdiag), g32(v), g32(f), g32(rest), g32(fric)
    Jg = g32(Je) if with_joint else None
    dt = 1.0 / 30
    out = solve_dynamics(B, nb, maxc, 3 if with_joint else 0, cb.count, Mg, vg, fg, rg, cg, cb, Jg, dt)
    torch.cuda.synchronize()
    got = out["v_new"].double().cpu().numpy()
    d64 = lambda t: t.double().cpu().numpy()
    worst = 0.0
    for k in range(B):
        n = counts[k]
        # the oracle sees the same fp32-rounded inputs and the same (fp32-rounded) contact frame as the kernel
        cs = [((d64(cb.c_n[k, q]), d64(cb.c_p1[k, q]), d64(cb.c_p2[k, q]), 0.0), int(cb.c_i1[k, q]), int(cb.c_i2[k, q])) for q in range(n)]
        ref = W.solve_dynamics(d64(Mg[k]), d64(vg[k]), d64(fg[k]), dt, cs, d64(rg[k]), d64(cg[k]),
                               d64(Jg[k]) if with_joint else None)
        scale = max(1.0, np.abs(ref).max())
        err = np.abs(got[k] - ref).max() / scale
        worst = max(worst, err)
        assert err <= 1e-4, (k, n, err)
        assert int(out["status"][k]) & 8 == 0
        # padded slots of z / s are reported as 0
        z = out["z"][k].cpu().numpy().reshape(-1)
        slots = np.concatenate([np.arange(n, maxc), maxc + np.arange(2 * n, 2 * maxc), 3 * maxc + np.arange(n, maxc)])
        assert np.all(z[slots] == 0)
    print("worst scaled new_v error", worst)


def _world_of(rec, B, k=0, **kw):
    from lcp_physics_amd.physics.batched_world import ContactWorld
    geom = _geom([shapes_of(rec)] * B)
    rep = lambda a, dt_: torch.tensor(np.broadcast_to(a, (B,) + a.shape).copy(), dtype=dt_, device=DEV)
    return ContactWorld(geom, rep(rec["p"][k], torch.float64), rep(rec["v"][k], torch.float32),
                        rep(rec["Mdiag"], torch.float32), rep(rec["f"], torch.float32), rep(rec["rest"], torch.float32),
                        rep(rec["fric"], torch.float32), Je=rep(rec["Je"], torch.float32), dt=float(rec["dt"]),
                        eps=float(rec["eps"]), tol=float(rec["tol"]), strict_no_penetration=bool(rec["strict"]), maxc=8, **kw)


@pytest.mark.parametrize("name", sorted(n for n, r in load_world_traj().items() if bool(r["post_stab"])))
def test_contact_world_post_stabilization_steps_match_reference(name):
    """ContactWorld(post_stab=True): every step of the reference trajectories recorded with post-stabilisation on
    (world.py:109-121, engines.py:80-116), each started from the reference's own state (pose, velocities, contacts
    re-detected at that pose).  Step by step because ten PDIPM iterations do not converge on the frictionless LCP of a
    resting contact (rhs ~ 0) and its output then amplifies rounding by ~1e8 (tests/test_world_oracle.py): a free run
    drifts by the 1e-5 the fp32 velocities put in.  Checked per step: dt accepted, contact count after the step, pose
    and velocities to 2e-4, the engine's dp to 2e-3 + 1e-3 |dp|."""
    rec = load_world_traj()[name]
    B = 5
    rep = lambda a, dt_: torch.tensor(np.broadcast_to(a, (B,) + a.shape).copy(), dtype=dt_, device=DEV)
    world = _world_of(rec, B, post_stab=True)
    from lcp_physics_amd.physics.contacts import find_contacts
    worst_p = worst_v = worst_dp = 0.0
    for k in range(1, len(rec["t"])):
        world.p.copy_(rep(rec["p"][k - 1], torch.float64))
        world.v.copy_(rep(rec["v"][k - 1], torch.float32))
        world.t.fill_(float(rec["t"][k - 1]))
        find_contacts(world.geom, world.p, maxc=world.maxc, eps=world.eps, out=world.contacts)
        assert world.contacts.count.cpu().tolist() == [int(rec["ncontacts"][k - 1])] * B, (name, k, "contacts before")
        out = world.step()
        t = world.t.cpu().numpy()
        assert np.abs(t - rec["t"][k]).max() < 1e-12, (name, k, "t", t, rec["t"][k])
        assert world.contacts.count.cpu().tolist() == [int(rec["ncontacts"][k])] * B, (name, k, "contact count")
        ep = np.abs(world.p.cpu().numpy() - rec["p"][k]).max()
        ev = np.abs(world.v.double().cpu().numpy() - rec["v"][k]).max()
        dp = out["post_stab"]["dp"].double().cpu().numpy()
        edp = (np.abs(dp - rec["dp"][k - 1]) - 1e-3 * np.abs(rec["dp"][k - 1])).max()
        worst_p, worst_v, worst_dp = max(worst_p, ep), max(worst_v, ev), max(worst_dp, edp)
        assert ep <= 2e-4 and ev <= 2e-4 and edp <= 2e-3, (name, k, ep, ev, edp)
    print(name, "worst |dp pose|", worst_p, "worst |dv|", worst_v, "worst post-stab dp error", worst_dp)


@pytest.mark.parametrize("name", ["stack3", "mixed", "stack3_poststab"])
def test_contact_world_graph_replay_is_bitwise_the_eager_run(name):
    """ContactWorld.run(n, graph=True) replays pairs of steps from a captured HIP graph: same kernels in the same order on
    the same buffers, so poses, velocities, times and contact lists must equal the eager run bit for bit."""
    rec = load_world_traj()[name]
    B, n = 6, 21                                                          # odd: two warm-up steps, nine replays, one eager tail
    a = _world_of(rec, B, post_stab=bool(rec["post_stab"]))
    b = _world_of(rec, B, post_stab=bool(rec["post_stab"]))
    for _ in range(n):
        a.step()
    b.run(n, graph=True)
    assert len(b._graphs) == 1
    b.run(4, graph=True)                                                  # odd phase now: a second graph is captured
    for _ in range(4):
        a.step()
    torch.cuda.synchronize()
    assert len(b._graphs) == 2
    assert torch.equal(a.p, b.p) and torch.equal(a.v, b.v) and torch.equal(a.t, b.t)
    assert torch.equal(a.contacts.count, b.contacts.count) and torch.equal(a.contacts.c_p1, b.contacts.c_p1)
    assert float(a.t.min()) > 0.3


def test_graphs_captured_before_a_differentiable_step_are_not_replayed_after_it():
    """run(graph) -> step(differentiable=True) -> run(graph) -> backward.  The differentiable step retires the contact buffers into
    its backward and replaces the state tensors; a graph captured earlier holds raw pointers to all of them.  The second run() must
    capture again: it advances the VISIBLE state exactly as eager steps do, and the gradient of the differentiable step is the one
    computed without any run() after it (a stale replay used to overwrite the contact records saved for the backward)."""
    rec = load_world_traj()["stack3"]
    B = 4

    def play(second_run):
        w = _world_of(rec, B)
        w.run(6, graph=True)
        w.f = w.f.clone().requires_grad_(True)
        out = w.step(differentiable=True)
        v_diff, p_diff = w.v, w.p
        assert not w._graphs                                               # nothing captured on the retired buffers may replay
        second_run(w)
        torch.cuda.synchronize()
        (v_diff.double().pow(2).sum() + p_diff.pow(2).sum()).backward()
        return w, w.f.grad.clone()

    a, ga = play(lambda w: w.run(6, graph=True))
    b, gb = play(lambda w: [w.step() for _ in range(6)])
    c, gc = play(lambda w: None)
    assert torch.equal(a.p, b.p) and torch.equal(a.v, b.v) and torch.equal(a.t, b.t)
    assert torch.equal(a.contacts.count, b.contacts.count)
    assert bool(torch.isfinite(ga).all()) and float(ga.abs().max()) > 0
    assert torch.equal(ga, gc) and torch.equal(gb, gc)


def test_post_stabilization_matches_oracle_with_ragged_counts():
    """`lcp_post_stabilization_f32` on stack scenes with random velocities and per-scene contact counts (0 = the direct
    KKT solve, engines.py:92-103) against oracle/pdipm_oracle.post_stabilization per scene, and the correction move
    p + (dp / 2) dt_k (world.py:110-117)."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics.batched_world import post_stabilization
    from lcp_physics_amd.physics.contacts import ContactBuffers
    from oracle import pdipm_oracle as O
    for nbox, pts in ((2, 2), (4, 4), (6, 4)):
        B = 12
        sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=70 + nbox, dtype=torch.float32)
        scg = sc.to(device=DEV)
        cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
        cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
        counts = [sc.nc, sc.nc, sc.nc - 1, sc.nc // 2, 1, 0, sc.nc, 3, 2, sc.nc, 0, sc.nc]
        count = torch.tensor(counts, dtype=torch.int32, device=DEV)
        g = torch.Generator().manual_seed(5)
        p = (torch.randn(B, sc.nb, 3, generator=g, dtype=torch.float64) * 100).to(DEV)
        dts = (torch.rand(B, generator=g, dtype=torch.float64) / 30).to(DEV)
        p_out = torch.empty_like(p)
        out = post_stabilization(B, sc.nb, sc.nc, 3, count, scg.Mdiag, scg.v, scg.rest, cb, scg.Je, p=p, dt_scene=dts,
                                 p_out=p_out)
        torch.cuda.synchronize()
        dp = out["dp"].double().cpu()
        # (the move uses the fp64 dp, `dp` is its fp32 rounding)
        moved = (p + (out["dp"].double() * 0.5) * dts[:, None, None]).cpu()
        assert float((p_out.cpu() - moved).abs().max()) <= 1e-7 * float(dp.abs().max()) / 30 + 1e-12
        worst = 0.0
        for k, n in enumerate(counts):
            d = lambda t: t[k:k + 1].double().cpu()
            if n == 0:
                ref = torch.tensor(W.post_stabilization(d(sc.Mdiag)[0].numpy(), d(sc.v)[0].numpy(), [], d(sc.rest)[0].numpy(),
                                                        d(sc.Je)[0].numpy()))
            else:
                ref = O.post_stabilization(d(sc.Mdiag), d(sc.v), d(sc.c_n)[:, :n], d(sc.c_p1)[:, :n], d(sc.c_p2)[:, :n],
                                           sc.c_i1[k:k + 1, :n], sc.c_i2[k:k + 1, :n], d(sc.rest), d(sc.Je))[0][0]
            err = float((dp[k] - ref).abs().max()) / max(1.0, float(ref.abs().max()))
            worst = max(worst, err)
            assert err <= 1e-4, (nbox, k, n, err)
            assert int(out["status"][k]) & 8 == 0
        print("post-stabilisation", nbox, pts, "worst scaled dp error", worst)


@pytest.mark.parametrize("name", sorted(n for n, r in load_world_traj().items() if not bool(r["post_stab"])))
def test_contact_world_follows_reference_trajectory(name):
    """ContactWorld.step() against trajectories of the unmodified reference World (world.py:72-122).
    Velocities are carried in fp32 between the two kernels, so poses (coordinates of ~500) and velocities
    (~100) are held to 2e-4 absolute over the whole run (measured: 4e-5); accepted step times (to 1e-12) and
    contact counts must agree exactly, i.e. every dt-halving decision of the reference is reproduced."""
    from lcp_physics_amd.physics.batched_world import ContactWorld
    rec = load_world_traj()[name]
    shapes = shapes_of(rec)
    B = 5                                                        # replicas (a wave holds 4 scenes: exercises the tail too)
    geom = _geom([shapes] * B)
    rep = lambda a, dt_: torch.tensor(np.broadcast_to(a, (B,) + a.shape).copy(), dtype=dt_, device=DEV)
    nb = len(shapes)
    if rec["no_contact"].size:                                  # Body.add_no_contact (bodies.py:104-106)
        nocon = torch.zeros(B, nb, nb, dtype=torch.uint8, device=DEV)
        for i, j in rec["no_contact"].tolist():
            nocon[:, i, j] = 1
        geom.no_contact = nocon
    # joints: a revolute `Joint` / `FixedJoint` needs the Jacobian that follows the pose (constraints.py:13-92) - a JointSet;
    # the constant-Jacobian records keep handing over the recorded Je
    kw = {}
    if any(int(x) in (1, 2) for x in rec["jtype"]):
        from lcp_physics_amd.physics.joints import JointSet
        kw["joints"] = JointSet.from_arrays(rec["jtype"], rec["jb1"], rec["jb2"], rec["jr1"], rec["jrot1"], B).to(DEV)
    else:
        kw["Je"] = rep(rec["Je"], torch.float32)
    # forces: a time-dependent ExternalForce (forces.py:14-18,29-48) is replayed from the recorded f(t) of every step
    f_t = torch.tensor(rec["f_t"], dtype=torch.float32, device=DEV)
    if float(np.abs(rec["f_t"] - rec["f_t"][0]).max()) > 0:
        step_of = {"k": 0}
        kw["force_fn"] = lambda t: f_t[step_of["k"]].unsqueeze(0).expand(B, -1, -1)
    world = ContactWorld(geom, rep(rec["p"][0], torch.float64), rep(rec["v"][0], torch.float32),
                         rep(rec["Mdiag"], torch.float32), rep(rec["f"], torch.float32), rep(rec["rest"], torch.float32),
                         rep(rec["fric"], torch.float32), dt=float(rec["dt"]),
                         eps=float(rec["eps"]), tol=float(rec["tol"]), strict_no_penetration=bool(rec["strict"]), maxc=8, **kw)
    assert world.contacts.count.cpu().tolist() == [int(rec["ncontacts"][0])] * B
    worst_p = worst_v = 0.0
    for k in range(1, len(rec["t"])):
        if "force_fn" in kw:
            step_of["k"] = k - 1
        if "joints" in kw:                                       # World.Je() at the pose the step starts from
            assert float((world.Je[0].double().cpu() - torch.tensor(rec["Je_t"][k - 1])).abs().max()) < 1e-4, (name, k, "Je")
        world.step()
        t = world.t.cpu().numpy()
        assert np.abs(t - rec["t"][k]).max() < 1e-12, (name, k, "t", t, rec["t"][k])
        assert world.contacts.count.cpu().tolist() == [int(rec["ncontacts"][k])] * B, (name, k, "contact count")
        ep = np.abs(world.p.cpu().numpy() - rec["p"][k]).max()
        ev = np.abs(world.v.double().cpu().numpy() - rec["v"][k]).max()
        worst_p, worst_v = max(worst_p, ep), max(worst_v, ev)
        assert ep <= 2e-4 and ev <= 2e-4, (name, k, ep, ev)
    print(name, "worst |dp|", worst_p, "worst |dv|", worst_v)


@pytest.mark.parametrize("nbox,maxc", [(6, 24), (6, 16), (5, 16)])
def test_contact_world_beyond_the_quad_sizes_follows_oracle(nbox, maxc):
    """6-7 bodies (nz = 18 / 21 > 16): contact kernel with one wave per scene + the solver with per-scene contact counts -
    the 32-contact class of lcp_big.hip (capacity 24) and the two-halves instantiation of lcp_quad.hip (capacity 16);
    a few steps against oracle/world_oracle.py (pinned on the reference World, tests/test_world_oracle.py)."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics.batched_world import ContactWorld
    B, nsteps = 3, 8
    w = scenes.make_drop_world(B, nbox=nbox, box=30.0, seed=11, gap=(0.2, 0.6))
    geom = _geom([w["shapes"]] * B)
    g = lambda k: w[k].to(DEV)
    world = ContactWorld(geom, g("p"), g("v"), g("Mdiag"), g("f"), g("rest"), g("fric"), Je=g("Je"), maxc=maxc)
    refs = []
    for s in range(B):
        d = lambda k: w[k][s].double().numpy()
        p, v = d("p"), d("v")
        cs = C.find_contacts(W.bodies_at(w["shapes"], p), eps=0.1)
        t = 0.0
        for _ in range(nsteps):
            p, v, cs, dt_used, _ = W.step_dt(w["shapes"], p, v, cs, d("Mdiag"), d("f"), d("rest"), d("fric"), d("Je"), 1.0 / 30)
            t += dt_used
        refs.append((p, v, len(cs), t))
    for _ in range(nsteps):
        world.step()
    world.check_capacity()
    for s in range(B):
        p, v, n, t = refs[s]
        assert abs(float(world.t[s]) - t) < 1e-12 and int(world.contacts.count[s]) == n, (s, float(world.t[s]), t, int(world.contacts.count[s]), n)
        assert np.abs(world.p[s].cpu().numpy() - p).max() < 2e-4 and np.abs(world.v[s].double().cpu().numpy() - v).max() < 2e-3, s


def test_mid_size_scenes_match_generic_and_oracle():
    """6-10 bodies with <= 16 contacts (the nz <= 32 instantiation of lcp_quad.hip) and 7 bodies / 24 contacts (the
    32-contact class of lcp_big.hip): new_v against the generic kernels and the fp64 oracle, with ragged counts."""
    from lcp_physics_amd import _lib, scenes
    from lcp_physics_amd.physics.batched_world import solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    from oracle import pdipm_oracle as O
    # nz 18 / 27 / 30 with 10 / 16 / 9 contacts (quad, two x-halves); 24 contacts (lcp_big, 32-contact class);
    # 12 bodies / 11 contacts: nz 36 > 32 (lcp_big, 16-contact class)
    for nbox, pts in ((5, 2), (8, 2), (9, 1), (6, 4), (11, 1)):
        B = 10
        sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=40 + nbox, dtype=torch.float32)
        scg = sc.to(device=DEV)
        cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
        cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
        counts = [sc.nc, sc.nc, sc.nc - 2, sc.nc // 2, 1, 0, sc.nc, 3, sc.nc - 1, sc.nc]
        count = torch.tensor(counts, dtype=torch.int32, device=DEV)
        run = lambda: solve_dynamics(B, sc.nb, sc.nc, 3, count, scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, scg.Je, sc.dt)
        got = run()["v_new"].double().cpu()
        _lib.set_path("generic")
        try:
            gen = run()["v_new"].double().cpu()
        finally:
            _lib.set_path("auto")
        assert float((got - gen).abs().max()) <= 1e-5 * max(1.0, float(gen.abs().max())), (nbox, float((got - gen).abs().max()))
        k = 0                                               # full contact list: against the oracle
        lcp64 = [None if t is None else t[k:k + 1].double() for t in O.assemble_lcp(*sc.assembly_args())]
        ref = -O.lcp_forward(*lcp64).x.reshape(sc.nb, 3)
        assert float((got[k] - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_config5_pile_solve_dynamics_matches_oracle():
    """BASELINE config 5 shape (11 bodies, 64 contacts, nineq 256) through lcp_solve_dynamics_f32 without hints = the
    one-wave-per-scene body-space kernel `lcp_primal_kernel<40>` (lcp_primal.hip; the contact-space `lcp_big.hip` is the A/B
    partner behind path "big", tests/test_hip_primal.py), with ragged contact counts: new_v within 1e-4 (scaled) of the fp64
    oracle on the same fp32 inputs, and the generic kernels' answer to 1e-5.  The pinned instantiation at the BASELINE batch is
    gated in tests/test_hip_headline_parity.py (configs4_4096x64_pile)."""
    from lcp_physics_amd import _lib, scenes
    from lcp_physics_amd.physics.batched_world import solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    from oracle import pdipm_oracle as O
    B = 12
    sc = scenes.make_pile_scenes(B=B, seed=21, dtype=torch.float32)
    scg = sc.to(device=DEV)
    cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    counts = [64, 64, 48, 33, 17, 64, 5, 0, 64, 20, 64, 1]
    count = torch.tensor(counts, dtype=torch.int32, device=DEV)
    run = lambda: solve_dynamics(B, sc.nb, sc.nc, 3, count, scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, scg.Je, sc.dt)
    out = run()
    torch.cuda.synchronize()
    got = out["v_new"].double().cpu()
    _lib.set_path("generic")
    try:
        ref_generic = run()["v_new"].double().cpu()
    finally:
        _lib.set_path("auto")
    worst = worst_ex = 0.0
    rows_masked = rows_total = 0
    for k in range(B):
        n = counts[k]
        one = lambda t: t[k:k + 1]
        if n > 0:
            args = (one(sc.Mdiag), one(sc.v), one(sc.f), sc.dt, sc.c_n[k:k + 1, :n], sc.c_p1[k:k + 1, :n], sc.c_p2[k:k + 1, :n],
                    sc.c_i1[k:k + 1, :n], sc.c_i2[k:k + 1, :n], one(sc.rest), one(sc.fric), one(sc.Je))
            lcp64 = [None if t is None else t.double() for t in O.assemble_lcp(*args)]
            rs = O.lcp_forward(*lcp64)
            ref = -rs.x.reshape(sc.nb, 3)
            # SURVEY 8d's metric for this config too: err_x scaled by the free motion, and the contact index sets
            # {i: z_i > s_i} of the normal, friction and cone rows, bit-exact where the oracle's decision is not a tie
            ex = float(parity.err_x(-got[k].reshape(1, -1), rs.x, lcp64[0], lcp64[1]).max())
            worst_ex = max(worst_ex, ex)
            assert ex <= 1e-4, (k, n, "err_x", ex)
            nc = sc.nc
            unpad = lambda t: torch.cat([t[k, :n], t[k, nc:nc + 2 * n], t[k, 3 * nc:3 * nc + n]]).double().cpu().reshape(1, -1)
            zg, sg = unpad(out["z"]), unpad(out["s"])
            dec = parity.decisive_rows(rs.z, rs.s)
            same = (parity.active_sets(zg, sg) == parity.active_sets(rs.z, rs.s)) | ~dec
            assert bool(same.all()), (k, n, "index sets", torch.nonzero(~same)[:8].tolist())
            rows_masked += int((~dec).sum()); rows_total += dec.numel()
        else:
            ref = torch.tensor(W.solve_dynamics(sc.Mdiag[k].numpy(), sc.v[k].numpy(), sc.f[k].numpy(), sc.dt, [], sc.rest[k].numpy(),
                                                sc.fric[k].numpy(), sc.Je[k].numpy()))
        scale = max(1.0, float(ref.abs().max()))
        err = float((got[k] - ref).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 1e-4, (k, n, err)
        assert float((got[k] - ref_generic[k]).abs().max()) / scale <= 1e-5, (k, n, "vs generic")
        assert int(out["status"][k]) & 8 == 0
    print("config 5 worst scaled error", worst, "worst err_x", worst_ex, "index-set rows masked", rows_masked, "of", rows_total)
    assert rows_masked <= 0.1 * rows_total
    # the same piles without the joint (a heavy free floor, neq = 0): big kernel against the generic kernels
    Md = scg.Mdiag.clone(); Md[:, 0] *= 1e4
    f0 = scg.f.clone(); f0[:, 0] = 0
    run0 = lambda: solve_dynamics(B, sc.nb, sc.nc, 0, count, Md, scg.v, f0, scg.rest, scg.fric, cb, None, sc.dt)
    a = run0()["v_new"].double().cpu()
    _lib.set_path("generic")
    try:
        g = run0()["v_new"].double().cpu()
    finally:
        _lib.set_path("auto")
    assert float((a - g).abs().max()) <= 1e-5 * max(1.0, float(g.abs().max())), float((a - g).abs().max())


@pytest.mark.parametrize("kind", ["stack", "pile"])
def test_batched_world_step_fixed_contacts(kind):
    """`BatchedWorld.step()` (fixed contact list): new_v = -x and p += v dt against the oracle, on the quad sizes
    (one fused launch) and on config-5 sized piles (lcp_solve_dynamics_f32 -> lcp_big.hip, then the integrator)."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics.batched_world import BatchedWorld
    from oracle import pdipm_oracle as O
    B = 6
    sc = (scenes.make_stack_scenes(B=B, nbox=3, pts_per_interface=2, seed=4, dtype=torch.float32) if kind == "stack"
          else scenes.make_pile_scenes(B=B, seed=4, dtype=torch.float32))
    lcp64 = [None if t is None else t.double() for t in O.assemble_lcp(*sc.assembly_args())]
    v_ref = -O.lcp_forward(*lcp64).x.reshape(B, sc.nb, 3)
    p_ref = sc.p.double() + v_ref * sc.dt
    world = BatchedWorld(sc.to(device=DEV))
    world.step()
    torch.cuda.synchronize()
    scale = v_ref.abs().reshape(B, -1).max(dim=1)[0].clamp_min(1.0).reshape(B, 1, 1)
    assert float(((world.get_v().double().cpu() - v_ref).abs() / scale).max()) < 1e-4
    assert float((world.get_p().double().cpu() - p_ref).abs().max()) < 1e-3            # fp32 poses of ~500
    assert abs(world.t - sc.dt) < 1e-12


def test_contact_world_refuses_initial_penetration():
    from lcp_physics_amd.physics.batched_world import ContactWorld
    shapes = [("rect", (500.0, 10.0)), ("rect", (40.0, 40.0))]
    geom = _geom([shapes])
    p = torch.tensor([[[0.0, 300.0, 400.0], [0.0, 300.0, 376.0]]], dtype=torch.float64, device=DEV)   # 1 px inside
    z = lambda *s: torch.zeros(*s, device=DEV)
    with pytest.raises(AssertionError):
        ContactWorld(geom, p, z(1, 2, 3), torch.ones(1, 2, 3, device=DEV), z(1, 2, 3), z(1, 2), z(1, 2))


def test_singular_pivot_scenes_of_a_settled_world_match_the_oracle():
    """The scenes of a settled `ContactWorld` that end a solve with LCP_ST_SINGULAR_T (bit 4: an exact zero pivot once
    s/z underflows the diagonal of T - a third of the batch once the stacks rest, profiles/r01_bench_world.json) are
    compared SEPARATELY against the oracle: the kernel then returns its best iterate, the twin of the reference's
    `except: return best` (pdipm.py:99-102), and that iterate has to be the oracle's answer for the same scene state
    (new_v to 1e-4 of the free motion, contact index sets identical where the oracle's decision is not a tie).
    The exact zero pivot is a property of the contact-space matrix T: the test forces that formulation (`set_path("big")`);
    the body-space variant the contact-list path runs by default factors another matrix and does not raise the bit."""
    from lcp_physics_amd import _lib, scenes
    from lcp_physics_amd.physics import batched_world as bw
    from lcp_physics_amd.physics import contacts as ct
    from oracle import pdipm_oracle as O
    B = 256
    w = scenes.make_drop_world(B, nbox=4, box=40.0)
    geom = ct.GeometryBatch.from_shapes(w["shapes"], B).to(DEV)
    g = lambda k: w[k].to(DEV)
    world = bw.ContactWorld(geom, g("p"), g("v"), g("Mdiag"), g("f"), g("rest"), g("fric"), Je=g("Je"), maxc=16)
    flagged_total = checked = 0
    _lib.set_path("big")
    for step in range(70):
        snap = None
        if step >= 40 and step % 6 == 0:                   # the scene state ENTERING the solve
            cb = world.contacts
            snap = {k: getattr(cb, k).clone() for k in ("c_n", "c_p1", "c_p2", "c_i1", "c_i2", "count")}
            snap["v"] = world.v.clone()
        out = world.step()
        if snap is None:
            continue
        st = out["status"].cpu()
        flagged = torch.nonzero((st & 4) != 0).flatten().tolist()
        assert int((st & ~4).max()) == 0                   # nothing but bit 4 is ever raised here
        flagged_total += len(flagged)
        cpu = lambda t: t.cpu()
        for k in flagged[:6]:
            n = int(snap["count"][k])
            assert 0 < n <= 16
            one = lambda t: cpu(t[k:k + 1])
            args = (one(world.Mdiag), one(snap["v"]), one(world.f), world.dt, one(snap["c_n"])[:, :n], one(snap["c_p1"])[:, :n],
                    one(snap["c_p2"])[:, :n], one(snap["c_i1"])[:, :n], one(snap["c_i2"])[:, :n], one(world.rest), one(world.fric),
                    one(world.Je))
            lcp64 = [None if t is None else t.double() for t in O.assemble_lcp(*args)]
            rs = O.lcp_forward(*lcp64)
            ex = float(parity.err_x(-out["v_new"][k].double().cpu().reshape(1, -1), rs.x, lcp64[0], lcp64[1]).max())
            assert ex <= 1e-4, (step, k, n, ex)
            unpad = lambda t: torch.cat([t[k, :n], t[k, 16:16 + 2 * n], t[k, 48:48 + n]]).double().cpu().reshape(1, -1)
            zg, sg = unpad(out["z"]), unpad(out["s"])
            dec = parity.decisive_rows(rs.z, rs.s)
            same = (parity.active_sets(zg, sg) == parity.active_sets(rs.z, rs.s)) | ~dec
            assert bool(same.all()), (step, k, "index sets", torch.nonzero(~same)[:8].tolist())
            checked += 1
    _lib.set_path("auto")
    print("scenes with status bit 4 over the sampled steps:", flagged_total, "checked against the oracle:", checked)
    assert checked >= 6, "the settled world no longer produces singular-pivot scenes: drop or re-seed this test"


def _rollout_world(d, rep, requires_grad=True):
    from lcp_physics_amd.physics.batched_world import ContactWorld
    from lcp_physics_amd.physics.contacts import GeometryBatch
    nv = d["force0"].shape[0]
    B = nv * rep
    rp = lambda a, dt_: torch.tensor(np.repeat(a, rep, axis=0), dtype=dt_, device=DEV)
    nb = d["rad"].shape[1]
    geom = GeometryBatch.from_shapes([("circle", float(r)) for r in d["rad"][0]], B)
    nocon = torch.zeros(B, nb, nb, dtype=torch.uint8)
    for i, j in d["no_contact"].tolist():
        nocon[:, i, j] = nocon[:, j, i] = 1
    geom.no_contact = nocon
    geom = geom.to(DEV)
    force0 = rp(d["force0"], torch.float32).requires_grad_(requires_grad)
    mult, t_push, pushed = float(d["mult"]), float(d["t_push"]), int(d["pushed_body"])

    def force_fn(t):                                    # ExternalForce(lambda t: force0 if t < 0.1 else ZEROS, multiplier) on body `pushed`
        on = (t < t_push).to(torch.float32).unsqueeze(1)
        z = torch.zeros(B, 1, 3, dtype=torch.float32, device=DEV)
        parts = [z] * nb
        parts[pushed] = (force0 * mult * on).unsqueeze(1)
        return torch.cat(parts, dim=1)

    world = ContactWorld(geom, rp(d["p0"], torch.float64), rp(d["v0"], torch.float32), rp(d["Mdiag"], torch.float32),
                         torch.zeros(B, nb, 3, device=DEV), rp(d["rest"], torch.float32), rp(d["fric"], torch.float32), Je=None,
                         dt=float(d["dt"]), maxc=2, force_fn=force_fn)
    return world, force0


def test_differentiable_steps_release_their_workspaces():
    """Every differentiable step keeps a 58 KB-per-scene workspace for its backward.  Until round 4 the autograd nodes held the tensor
    they return (`ctx.out["v_new"]`, `ctx.sol.x`): a reference cycle through C++ that Python's collector cannot see, so NO roll-out was
    ever freed (8 GB per 36-step roll-out of 4096 scenes).  Three roll-outs + backward, then the same through the dense LCPFunction:
    the memory in use afterwards is what it was after the first."""
    import gc
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import LCPFunction
    from lcp_physics_amd.physics import assemble_contacts
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "rollout_grad.npz"))
    world, force0 = _rollout_world(d, 32)
    p0, v0 = world.p.clone(), world.v.clone()

    def rollout():
        force0.grad = None
        world.restart(p0, v0)
        for _ in range(6):
            world.step(differentiable=True)
        world.p.sum().backward()

    used = []
    for _ in range(4):
        rollout()
        torch.cuda.synchronize()
        gc.collect()
        used.append(torch.cuda.memory_allocated())
    assert used[-1] <= used[0] + (1 << 20), used
    sc = scenes.make_stack_scenes(B=64, nbox=4, pts_per_interface=4, seed=3, dtype=torch.float32).to(device=DEV)
    lcp = [t.clone().requires_grad_(i == 1) for i, t in enumerate(assemble_contacts(sc))]
    used = []
    for _ in range(4):
        x = LCPFunction(check=False)(*lcp)
        x.sum().backward()
        del x
        torch.cuda.synchronize()
        gc.collect()
        used.append(torch.cuda.memory_allocated())
    assert used[-1] <= used[0] + (1 << 20), used


def test_contact_frame_backward_matches_autograd_of_the_circle_record():
    """`lcp_contact_frame_backward_f64` against torch autograd of the circle / circle contact tuple (contacts.py:68-79)."""
    from lcp_physics_amd.physics import contacts as ct
    B, nb = 64, 4
    g = torch.Generator().manual_seed(12)
    rad = 20.0 + 10.0 * torch.rand(nb, generator=g, dtype=torch.float64)
    geom = ct.GeometryBatch.from_shapes([("circle", float(r)) for r in rad], B).to(DEV)
    p = torch.zeros(B, nb, 3, dtype=torch.float64)
    p[:, :, 1] = torch.arange(nb, dtype=torch.float64) * 48.0 + 4.0 * torch.rand(B, nb, generator=g, dtype=torch.float64)
    p[:, :, 2] = 10.0 * torch.rand(B, nb, generator=g, dtype=torch.float64)
    pg = p.to(DEV)
    cb = ct.find_contacts(geom, pg, maxc=6, eps=30.0)
    cnt = cb.count.cpu()
    assert int(cnt.min()) >= 2
    gn, g1, g2 = [torch.randn(B, 6, 2, generator=g).to(DEV) for _ in range(3)]
    dp = ct.contact_frame_backward(geom, pg, cb, gn, g1, g2, eps=30.0).cpu()
    pt = p.clone().requires_grad_(True)
    loss = 0.0
    i1, i2 = cb.c_i1.cpu().long(), cb.c_i2.cpu().long()
    for b in range(B):
        for c in range(int(cnt[b])):
            a, o = int(i1[b, c]), int(i2[b, c])
            d = pt[b, a, 1:] - pt[b, o, 1:]
            dist = d.norm(); n = d / dist; pen = rad[a] + rad[o] - dist
            c1, c2 = -n * (rad[a] - pen / 2), n * (rad[o] - pen / 2)
            loss = loss + (n * gn[b, c].double().cpu()).sum() + (c1 * g1[b, c].double().cpu()).sum() + (c2 * g2[b, c].double().cpu()).sum()
            assert float((n.detach().float() - cb.c_n[b, c].cpu()).abs().max()) < 1e-6      # same record the kernel produced
    loss.backward()
    assert float((dp - pt.grad).abs().max()) <= 1e-12 * max(1.0, float(pt.grad.abs().max()))


def test_rollout_gradient_matches_the_reference_autograd():
    """A batched `grad_demo` (demos/grad_demo.py:19-83): B = 1024 scenes of three balls, 36 `ContactWorld.step(differentiable=
    True)` each, loss = |target - ball| after the roll-out, back-propagated to the force that pushed the first ball for
    0.1 s.  Final poses, losses and d(loss)/d(force) against what the UNMODIFIED reference produced for the same eight
    scenes by its own autograd (tests/golden/rollout_grad.npz, oracle/make_golden_rollout.py): 1e-4."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_grad.npz"))
    rep = 128
    world, force0 = _rollout_world(d, rep)
    nsteps = int(d["nsteps"])
    ncs = []
    for _ in range(nsteps):
        world.step(differentiable=True)
        ncs.append(world.contacts.count.clone())
    a, b = [int(i) for i in d["loss_bodies"]]
    pos = world.p[:, :, 1:]
    loss = (pos[:, a] - pos[:, b]).norm(dim=1)
    loss.sum().backward()
    torch.cuda.synchronize()
    # same trajectory: clocks (dt halving), contact counts, final poses
    assert np.abs(world.t.cpu().numpy()[::rep] - d["t"][:, -1]).max() < 1e-12
    assert (torch.stack(ncs, 1).cpu().numpy()[::rep] == d["ncontacts"]).all()
    pf = world.p.detach().cpu().numpy()[::rep]
    assert np.abs(pf - d["p_final"]).max() <= 1e-4, np.abs(pf - d["p_final"]).max()
    ls = loss.detach().cpu().numpy()[::rep]
    assert np.abs(ls - d["loss"]).max() <= 1e-5 * np.abs(d["loss"]).max()
    gr = force0.grad.cpu().numpy()
    assert np.abs(gr.reshape(-1, rep, 3) - gr[::rep][:, None]).max() == 0.0          # replicas are bitwise replicas
    ref = d["grad"]
    err = np.abs(gr[::rep] - ref).max(axis=1) / np.abs(ref).max(axis=1)
    print("roll-out gradient: worst relative error", err.max(), "per scene", np.array2string(err, precision=2))
    assert err.max() <= 1e-4, err


def test_rollout_gradient_through_hull_contacts_matches_the_reference_autograd():
    """The same, with contacts that involve hulls: a ball landing on a fixed floor (circle / hull, GJK), rolling into a box that
    slides and tips on the floor (hull / hull: SAT, incident edge, clipping), both pushed for 0.1 s by learnable forces, 40 steps,
    loss = |ball - box|.  d(loss)/d(both forces) - through `lcp_step_backward_f32` and the forward-mode contact-frame
    derivative `lcp_contact_frame_backward_f64` - against the unmodified reference's autograd on six scenes."""
    from lcp_physics_amd.physics.batched_world import ContactWorld
    from lcp_physics_amd.physics.contacts import GeometryBatch
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_grad.npz"))
    nv, rep = d["h_force_ball"].shape[0], 32
    B = nv * rep
    rp = lambda a, dt_: torch.tensor(np.repeat(a, rep, axis=0), dtype=dt_, device=DEV)
    shapes = [("circle", float(s[0])) if int(k) == 0 else ("rect", (float(s[0]), float(s[1]))) for k, s in zip(d["h_kind"][0], d["h_size"][0])]
    geom = GeometryBatch.from_shapes(shapes, B).to(DEV)
    fb = rp(d["h_force_ball"], torch.float32).requires_grad_(True)
    fx = rp(d["h_force_box"], torch.float32).requires_grad_(True)
    grav = rp(d["h_gravity"], torch.float32)
    mult, t_push = float(d["mult"]), float(d["t_push"])

    def force_fn(t):
        on = (t < t_push).to(torch.float32).unsqueeze(1)
        z = torch.zeros(B, 1, 3, dtype=torch.float32, device=DEV)
        return grav + torch.cat([z, (fb * mult * on).unsqueeze(1), (fx * mult * on).unsqueeze(1)], dim=1)

    world = ContactWorld(geom, rp(d["h_p0"], torch.float64), rp(d["h_v0"], torch.float32), rp(d["h_Mdiag"], torch.float32),
                         torch.zeros(B, 3, 3, device=DEV), rp(d["h_rest"], torch.float32), rp(d["h_fric"], torch.float32),
                         Je=rp(d["h_Je"], torch.float32), dt=float(d["dt"]), maxc=8, force_fn=force_fn)
    ncs = []
    for _ in range(int(d["h_nsteps"])):
        world.step(differentiable=True)
        ncs.append(world.contacts.count.clone())
    pos = world.p[:, :, 1:]
    loss = (pos[:, 1] - pos[:, 2]).norm(dim=1)
    loss.sum().backward()
    torch.cuda.synchronize()
    t_ok = np.abs(world.t.cpu().numpy()[::rep] - d["h_t"][:, -1]) < 1e-12
    n_ok = (torch.stack(ncs, 1).cpu().numpy()[::rep] == d["h_ncontacts"]).all(axis=1)
    same = t_ok & n_ok                                 # scenes whose every dt-halving / contact-count decision matched the reference's
    print("scenes on the reference's trajectory:", same.tolist())
    assert same.sum() >= nv - 1
    pf = world.p.detach().cpu().numpy()[::rep]
    assert np.abs(pf - d["h_p_final"])[same].max() <= 5e-4, np.abs(pf - d["h_p_final"])[same].max()
    gb, gx = fb.grad.cpu().numpy()[::rep], fx.grad.cpu().numpy()[::rep]
    ref = np.concatenate([d["h_grad_ball"], d["h_grad_box"]], axis=1)
    got = np.concatenate([gb, gx], axis=1)
    err = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
    print("hull roll-out gradient: relative error per scene", np.array2string(err, precision=2))
    if os.environ.get("LCP_TEST_VERBOSE"):
        print(np.array2string(got, precision=5)); print(np.array2string(ref, precision=5))
    assert err[same].max() <= 1e-5, err


@pytest.mark.parametrize("scene,with_dJe", [("j_", True), ("k_", True), ("j_", False), ("k_", False)])
def test_rollout_gradient_through_pose_dependent_joints_matches_the_reference_autograd(scene, with_dJe):
    """Joints whose Jacobian follows the pose.  "j_": a double pendulum (two revolute `Joint`s, one to the world) swinging into
    a free ball; "k_": a dumbbell (two discs welded by a `FixedJoint`) pushed, spinning, into a ball.  Learnable forces on the
    first body and on the ball, 36 steps, loss = |ball - second body|.  The reference differentiates through `Joint.J()` /
    `FixedJoint.J()` (pos1 = r1 (cos rot1, sin rot1), rot1 += body1.v[0] dt, pos2 = body1.pos + pos1 - body2.pos:
    constraints.py:26-85); here dL/dJe comes from `lcp_step_backward_je_f32` and reaches the pose and the joint angle through
    `JointSet.jacobian_torch`.  Against the unmodified reference's autograd on six scenes each; with the Jacobian held constant
    in the backward (`with_dJe=False`) the same comparison must FAIL - the test sees the path."""
    from lcp_physics_amd.physics.batched_world import ContactWorld
    from lcp_physics_amd.physics.contacts import GeometryBatch
    from lcp_physics_amd.physics.joints import JointSet
    d0 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_grad.npz"))
    d = {k[len(scene):]: d0[k] for k in d0.files if k.startswith(scene)}
    nv, rep = d["force_first"].shape[0], 32
    B = nv * rep
    rp = lambda a, dt_: torch.tensor(np.repeat(a, rep, axis=0), dtype=dt_, device=DEV)
    nb = d["rad"].shape[1]
    geom = GeometryBatch.from_shapes([("circle", float(r)) for r in d["rad"][0]], B)
    nocon = torch.zeros(B, nb, nb, dtype=torch.uint8)
    for i, j in d["no_contact"][0].tolist():
        nocon[:, i, j] = nocon[:, j, i] = 1
    geom.no_contact = nocon
    geom = geom.to(DEV)
    joints = JointSet.from_arrays(d["jtype"][0], d["jb1"][0], d["jb2"][0], d["jr1"][0], d["jrot1"][0], B).to(DEV)
    f1 = rp(d["force_first"], torch.float32).requires_grad_(True)
    f2 = rp(d["force_ball"], torch.float32).requires_grad_(True)
    grav = rp(d["gravity"], torch.float32)
    mult, t_push = float(d0["mult"]), float(d0["t_push"])

    def force_fn(t):
        on = (t < t_push).to(torch.float32).unsqueeze(1)
        z = torch.zeros(B, 1, 3, dtype=torch.float32, device=DEV)
        return grav + torch.cat([(f1 * mult * on).unsqueeze(1), z, (f2 * mult * on).unsqueeze(1)], dim=1)

    world = ContactWorld(geom, rp(d["p0"], torch.float64), rp(d["v0"], torch.float32), rp(d["Mdiag"], torch.float32),
                         torch.zeros(B, nb, 3, device=DEV), rp(d["rest"], torch.float32), rp(d["fric"], torch.float32),
                         joints=joints, dt=float(d0["dt"]), maxc=8, force_fn=force_fn)
    assert np.abs(world.Je.cpu().numpy()[::rep] - d["Je"]).max() <= 1e-5
    if not with_dJe:
        joints.__dict__["_pose_dep"] = False            # the round-1 behaviour: Je a constant of the backward
    ncs = []
    for _ in range(int(d["nsteps"])):
        world.step(differentiable=True)
        ncs.append(world.contacts.count.clone())
    pos = world.p[:, :, 1:]
    loss = (pos[:, 2] - pos[:, 1]).norm(dim=1)
    loss.sum().backward()
    torch.cuda.synchronize()
    t_ok = np.abs(world.t.cpu().numpy()[::rep] - d["t"][:, -1]) < 1e-12
    n_ok = (torch.stack(ncs, 1).cpu().numpy()[::rep] == d["ncontacts"]).all(axis=1)
    same = t_ok & n_ok
    print("scenes on the reference's trajectory:", same.tolist())
    assert same.sum() >= nv - 1
    pf = world.p.detach().cpu().numpy()[::rep]
    assert np.abs(pf - d["p_final"])[same].max() <= 2e-3, np.abs(pf - d["p_final"])[same].max()
    ref = np.concatenate([d["grad_first"], d["grad_ball"]], axis=1)
    got = np.concatenate([f1.grad.cpu().numpy()[::rep], f2.grad.cpu().numpy()[::rep]], axis=1)
    err = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
    print("jointed roll-out gradient (%s, dJe %s): relative error per scene" % (scene, with_dJe), np.array2string(err, precision=2))
    if os.environ.get("LCP_TEST_VERBOSE"):
        print(np.array2string(got, precision=5)); print(np.array2string(ref, precision=5))
    if with_dJe:
        assert err[same].max() <= 1e-4, err
    else:
        assert err[same].max() > 1e-2, err


@pytest.mark.parametrize("anchors_follow_p0", [True, False])
def test_rollout_gradient_reaches_the_initial_positions_through_the_joint_anchors(anchors_follow_p0):
    """The reference's `Joint.__init__` takes the anchor's polar coordinates from `pos - body1.pos` (constraints.py:21-23,
    utils.py:75-82): an initial position that requires grad reaches the loss through (r1, rot1) and every later `Joint.J()` as well as
    through the state.  "a_" scenes of the fixture: the double pendulum of "j_" with its bodies' initial positions as leaves; four
    scenes against the unmodified reference's autograd.  `JointSet.from_list(joints, p0)` keeps that graph; built from a detached
    `p0` (`anchors_follow_p0=False`) the same comparison must FAIL - the test sees the path."""
    from lcp_physics_amd.physics.batched_world import ContactWorld
    from lcp_physics_amd.physics.contacts import GeometryBatch
    from lcp_physics_amd.physics.joints import JointSet
    d0 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_grad.npz"))
    d = {k[2:]: d0[k] for k in d0.files if k.startswith("a_")}
    nv, rep = d["force_first"].shape[0], 16
    B = nv * rep
    rp = lambda a, dt_: torch.tensor(np.repeat(a, rep, axis=0), dtype=dt_, device=DEV)
    nb = d["rad"].shape[1]
    geom = GeometryBatch.from_shapes([("circle", float(r)) for r in d["rad"][0]], B)
    nocon = torch.zeros(B, nb, nb, dtype=torch.uint8)
    for i, j in d["no_contact"][0].tolist():
        nocon[:, i, j] = nocon[:, j, i] = 1
    geom.no_contact = nocon
    geom = geom.to(DEV)
    p0 = rp(d["p0"], torch.float64).requires_grad_(True)
    # (oracle/make_golden_rollout.py::make_world_pendulum: Joint(bob, None, [300, 220]), Joint(bob, link, [300, 330]))
    joints = JointSet.from_list([("joint", 0, None, (300.0, 220.0)), ("joint", 0, 1, (300.0, 330.0))], p0 if anchors_follow_p0 else p0.detach())
    assert np.abs(joints.jr1.detach().cpu().numpy()[::rep] - d["jr1"]).max() < 1e-12
    assert np.abs(joints.jrot1.detach().cpu().numpy()[::rep] - d["jrot1"]).max() < 1e-12
    f1, f2, grav = rp(d["force_first"], torch.float32), rp(d["force_ball"], torch.float32), rp(d["gravity"], torch.float32)
    mult, t_push = float(d0["mult"]), float(d0["t_push"])

    def force_fn(t):
        on = (t < t_push).to(torch.float32).unsqueeze(1)
        z = torch.zeros(B, 1, 3, dtype=torch.float32, device=DEV)
        return grav + torch.cat([(f1 * mult * on).unsqueeze(1), z, (f2 * mult * on).unsqueeze(1)], dim=1)

    world = ContactWorld(geom, p0, rp(d["v0"], torch.float32), rp(d["Mdiag"], torch.float32), torch.zeros(B, nb, 3, device=DEV),
                         rp(d["rest"], torch.float32), rp(d["fric"], torch.float32), joints=joints, dt=float(d0["dt"]), maxc=8, force_fn=force_fn)
    assert np.abs(world.Je.cpu().numpy()[::rep] - d["Je"]).max() <= 1e-5
    ncs = []
    for _ in range(int(d["nsteps"])):
        world.step(differentiable=True)
        ncs.append(world.contacts.count.clone())
    pos = world.p[:, :, 1:]
    (pos[:, 2] - pos[:, 1]).norm(dim=1).sum().backward()
    torch.cuda.synchronize()
    same = (np.abs(world.t.cpu().numpy()[::rep] - d["t"][:, -1]) < 1e-12) & (torch.stack(ncs, 1).cpu().numpy()[::rep] == d["ncontacts"]).all(axis=1)
    print("scenes on the reference's trajectory:", same.tolist())
    assert same.sum() >= nv - 1
    got, ref = p0.grad.cpu().numpy()[::rep, :, 1:].reshape(nv, -1), d["grad_p0"].reshape(nv, -1)
    err = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
    print("d(loss)/d(initial positions), anchors follow p0 = %s: relative error per scene" % anchors_follow_p0, np.array2string(err, precision=2))
    if anchors_follow_p0:
        assert err[same].max() <= 1e-4, err
    else:
        assert err[same].max() > 1e-2, err


@pytest.mark.parametrize("nbox,pts,extra_rows", [(2, 2, 0), (4, 2, 0), (6, 2, 0), (4, 2, 4), (8, 2, 5)])
def test_post_stabilization_backward_matches_oracle(nbox, pts, extra_rows):
    """`lcp_post_stabilization_backward_f32` against the fp64 oracle end to end: per scene, the oracle solves the frictionless LCP
    of engines.py:80-116, `lcp.py:37-64` gives the dense gradients, and autograd carries them through the assembly (ge = Je v,
    gc = Jc v (1 - restitutions), G = Jc, Q = M) to Mdiag, v, rest, the contact frame and Je.  Ragged contact counts (0 = the
    direct KKT solve); with `extra_rows` the scene has 3 + extra_rows equality rows (the 16-row instantiation).  Two contact
    points per interface and joints that leave no body pinned twice: the multipliers are unique, so the gradients with respect
    to the contact frame and Je are (four collinear points per interface leave z, hence dG, undetermined - SURVEY §8d)."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics.batched_world import post_stabilization, post_stabilization_backward
    from lcp_physics_amd.physics.contacts import ContactBuffers
    from oracle import pdipm_oracle as O
    B = 12
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=170 + nbox, dtype=torch.float32)
    if extra_rows:
        from tests.test_hip_primal import _with_joint_rows
        sc = _with_joint_rows(sc, 3 + extra_rows)
    e = sc.Je.shape[1]
    scg = sc.to(device=DEV)
    cb = ContactBuffers(B, sc.nb, sc.nc, DEV)
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
    counts = [sc.nc, sc.nc, sc.nc - 1, sc.nc // 2, 1, 0, sc.nc, 3, 2, sc.nc, 0, sc.nc]
    count = torch.tensor(counts, dtype=torch.int32, device=DEV)
    out = post_stabilization(B, sc.nb, sc.nc, e, count, scg.Mdiag, scg.v, scg.rest, cb, scg.Je)
    cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(4), dtype=torch.float32)
    g = post_stabilization_backward(B, sc.nb, sc.nc, e, scg.Mdiag, scg.v, scg.rest, cb, scg.Je, cot.to(DEV), out, want_Je=True)
    torch.cuda.synchronize()
    g = {k: t.double().cpu() for k, t in g.items()}
    worst = {}
    for k, n in enumerate(counts):
        leaf = lambda t: t[k:k + 1].double().clone().requires_grad_(True)
        Md, v, rest, Je = leaf(sc.Mdiag), leaf(sc.v), leaf(sc.rest), leaf(sc.Je)
        cn, cp1, cp2 = leaf(sc.c_n[:, :n]), leaf(sc.c_p1[:, :n]), leaf(sc.c_p2[:, :n])
        cx = -cot[k:k + 1].double().reshape(1, -1)                                  # dp = -x
        if n == 0:                                                                  # engines.py:92-103: x = P^-1 [0; ge]
            nz = 3 * sc.nb
            Pm = torch.cat([torch.cat([torch.diag(Md.reshape(-1)), -Je[0].t()], dim=1),
                            torch.cat([Je[0], torch.zeros(e, e, dtype=torch.float64)], dim=1)])
            x = torch.linalg.solve(Pm, torch.cat([torch.zeros(nz, dtype=torch.float64), Je[0] @ v.reshape(-1)]))[:nz]
            (x * cx[0]).sum().backward()
        else:
            lcp = O.assemble_post_stabilization(Md, v, cn, cp1, cp2, sc.c_i1[k:k + 1, :n], sc.c_i2[k:k + 1, :n], rest, Je)
            det = [None if t is None else t.detach() for t in lcp]
            sol = O.lcp_forward(*det)
            gr = O.lcp_backward(sol, *det, cx)
            outs, cots = [], []
            for t, key in zip(lcp, ("dQ", "dp", "dG", "dh", "dA", "db", "dF")):
                if t is not None and t.requires_grad and gr[key] is not None:
                    outs.append(t); cots.append(gr[key])
            torch.autograd.backward(outs, cots)
        ref = {"Mdiag": Md.grad, "v": v.grad, "rest": rest.grad, "Je": Je.grad}
        if n:
            ref.update({"c_n": cn.grad, "c_p1": cp1.grad, "c_p2": cp2.grad})
        for key, r in ref.items():
            r = torch.zeros_like(leaf(getattr(sc, key))) if r is None else r
            got = g[key][k:k + 1]
            if key in ("c_n", "c_p1", "c_p2"):
                assert float(got[:, n:].abs().max()) == 0.0 if n < sc.nc else True          # padded slots
                got = got[:, :n]
            scale = max(float(r.abs().max()), 1e-6 * max(float(ref["v"].abs().max()), 1e-30))
            err = float((got - r).abs().max()) / scale
            worst[key] = max(worst.get(key, 0.0), err)
            # (the contact-frame gradients are differences of products that nearly cancel, as in test_hip_step_backward.py)
            assert err <= (2e-3 if key in ("c_n", "c_p1", "c_p2") else 1e-4), (nbox, k, n, key, err)
    print("post-stabilisation backward", nbox, pts, e, {k: "%.1e" % v for k, v in worst.items()})


@pytest.mark.parametrize("scene", ["c_", "d_"])
def test_rollout_gradient_through_a_chain_with_post_stabilization_matches_the_reference_autograd(scene):
    """`experiments/inference.py:26-89`: a chain of links on revolute joints hit by a projectile, `World(post_stab=True)`; "c_": four
    links (8 equality rows, 30 steps, six scenes), "d_": the experiment's own ten links (20 equality rows, 11 bodies, 36 steps,
    three scenes).  The parameters are the links' mass (inertia, mass and gravity follow it) and the
    projectile's push.  d(loss)/d(mass) and d(loss)/d(push) through `SolveDynamicsFunction`, `PostStabilizationFunction`, the
    joints' Jacobians and the contact frames, against the unmodified reference's autograd on six scenes."""
    from lcp_physics_amd.physics.batched_world import ContactWorld
    from lcp_physics_amd.physics.contacts import GeometryBatch
    from lcp_physics_amd.physics.joints import JointSet
    d0 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_grad.npz"))
    d = {k[2:]: d0[k] for k in d0.files if k.startswith(scene)}
    nv, rep = d["mass"].shape[0], 16
    B = nv * rep
    rp = lambda a, dt_: torch.tensor(np.repeat(a, rep, axis=0), dtype=dt_, device=DEV)
    shapes = [("circle", float(s[0])) if int(k) == 0 else ("rect", (float(s[0]), float(s[1]))) for k, s in zip(d["kind"][0], d["size"][0])]
    nb = len(shapes)
    geom = GeometryBatch.from_shapes(shapes, B)
    nocon = torch.zeros(B, nb, nb, dtype=torch.uint8)
    for i, j in d["no_contact"][0].tolist():
        nocon[:, i, j] = nocon[:, j, i] = 1
    geom.no_contact = nocon
    geom = geom.to(DEV)
    joints = JointSet.from_arrays(d["jtype"][0], d["jb1"][0], d["jb2"][0], d["jr1"][0], d["jrot1"][0], B).to(DEV)
    mass = rp(d["mass"], torch.float32).requires_grad_(True)
    push = rp(d["force"], torch.float32).requires_grad_(True)
    per_mass, grav = rp(d["Mdiag_per_mass"], torch.float32), rp(d["gravity_per_mass"], torch.float32)
    fixed = rp(d["Mdiag"], torch.float32) * (per_mass == 0).to(torch.float32)          # the projectile's own mass matrix
    Mdiag = per_mass * mass.reshape(B, 1, 1) + fixed                                    # bodies.py:44-47,269-270 with mass = the parameter
    mult, t_push = float(d["mult"]), float(d0["t_push"])

    def force_fn(t):
        on = (t < t_push).to(torch.float32).reshape(B, 1, 1)
        sel = torch.zeros(1, nb, 1, dtype=torch.float32, device=DEV)
        sel[0, nb - 1, 0] = 1.0
        return grav * mass.reshape(B, 1, 1) + sel * (push * mult).unsqueeze(1) * on     # forces.py:51-67 Gravity, :29-48 ExternalForce

    world = ContactWorld(geom, rp(d["p0"], torch.float64), rp(d["v0"], torch.float32), Mdiag, torch.zeros(B, nb, 3, device=DEV),
                         rp(d["rest"], torch.float32), rp(d["fric"], torch.float32), joints=joints, dt=float(d0["dt"]), maxc=8,
                         force_fn=force_fn, post_stab=True)
    world.Mdiag = Mdiag                                                                 # (keep the graph: the constructor detaches nothing but copies)
    ncs = []
    for _ in range(int(d["nsteps"])):
        world.step(differentiable=True)
        ncs.append(world.contacts.count.clone())
    target = rp(d["target"], torch.float64)
    loss = ((world.p - target) ** 2).mean(dim=(1, 2))
    loss.sum().backward()
    torch.cuda.synchronize()
    t_ok = np.abs(world.t.cpu().numpy()[::rep] - d["t"][:, -1]) < 1e-12
    n_ok = (torch.stack(ncs, 1).cpu().numpy()[::rep] == d["ncontacts"]).all(axis=1)
    same = t_ok & n_ok
    print("scenes on the reference's trajectory:", same.tolist())
    assert same.sum() >= nv - 1 and same.sum() >= 2
    pf = world.p.detach().cpu().numpy()[::rep]
    assert np.abs(pf - d["p_final"])[same].max() <= 2e-3, np.abs(pf - d["p_final"])[same].max()
    ls = loss.detach().cpu().numpy()[::rep]
    assert (np.abs(ls - d["loss"]) / np.abs(d["loss"]))[same].max() <= 1e-5
    ref = np.concatenate([d["grad_mass"].reshape(nv, 1), d["grad_force"]], axis=1)
    got = np.concatenate([mass.grad.cpu().numpy()[::rep].reshape(nv, 1), push.grad.cpu().numpy()[::rep]], axis=1)
    err = np.abs(got - ref) / np.abs(ref).max(axis=1, keepdims=True)
    print("chain + post-stabilisation roll-out gradient (%s): relative error per scene" % scene, np.array2string(err.max(axis=1), precision=2))
    if os.environ.get("LCP_TEST_VERBOSE"):
        print(np.array2string(got, precision=5)); print(np.array2string(ref, precision=5))
    assert err[same].max() <= 1e-4, err
    # the same roll-out with the plain `step()` (joints + post-stabilisation, no autograd): every accepted dt and every contact
    # count of the reference's run, step by step, and its final poses
    with torch.no_grad():
        joints2 = JointSet.from_arrays(d["jtype"][0], d["jb1"][0], d["jb2"][0], d["jr1"][0], d["jrot1"][0], B).to(DEV)
        plain = ContactWorld(geom, rp(d["p0"], torch.float64), rp(d["v0"], torch.float32), Mdiag.detach(), torch.zeros(B, nb, 3, device=DEV),
                             rp(d["rest"], torch.float32), rp(d["fric"], torch.float32), joints=joints2, dt=float(d0["dt"]), maxc=8,
                             force_fn=force_fn, post_stab=True)
        on_track = np.ones(nv, dtype=bool)
        for k in range(int(d["nsteps"])):
            plain.step()
            on_track &= (np.abs(plain.t.cpu().numpy()[::rep] - d["t"][:, k]) < 1e-12) & (plain.contacts.count.cpu().numpy()[::rep] == d["ncontacts"][:, k])
        assert on_track.sum() >= nv - 1 and on_track.sum() >= 2, on_track.tolist()
        pf2 = plain.p.cpu().numpy()[::rep]
        assert np.abs(pf2 - d["p_final"])[on_track].max() <= 2e-3, np.abs(pf2 - d["p_final"])[on_track].max()


def test_mass_inference_through_differentiable_rollouts_recovers_the_mass():
    """`experiments/inference.py:26-89` end to end on a small batch (tools/experiments/mass_inference.py is the full-size run): eight
    chains of four links start from wrong masses, ten RMSprop iterations on log-mass through 24 differentiable steps with
    post-stabilisation bring every one of them within 1 % of the mass that produced the observed trajectory."""
    from lcp_physics_amd import scenes
    B, links, steps, true_mass = 8, 4, 24, 0.7

    def rollout(mass):
        world = scenes.make_chain_world(mass.shape[0], links=links, mass=mass, device=DEV)
        poses = []
        for _ in range(steps):
            world.step(differentiable=True)
            poses.append(world.p)
        return torch.stack(poses, 1)

    with torch.no_grad():
        observed = rollout(torch.full((1,), true_mass, device=DEV))
    log_m = torch.log(torch.linspace(0.35, 1.6, B)).to(DEV).requires_grad_(True)
    optim = torch.optim.RMSprop([log_m], lr=0.05)
    first = None
    for _ in range(10):
        optim.zero_grad()
        loss = ((rollout(log_m.exp()) - observed) ** 2).mean(dim=(1, 2, 3))
        first = loss.detach().clone() if first is None else first
        loss.sum().backward()
        optim.step()
    m = log_m.detach().exp().cpu()
    assert bool((loss.detach() < 1e-3 * first).all()), (first.tolist(), loss.tolist())
    assert float((m - true_mass).abs().max()) < 0.01 * true_mass, m.tolist()


@pytest.mark.parametrize("post_stab", [False, True])
def test_chain_world_graph_replay_is_bitwise_the_eager_run(post_stab):
    """The jointed world under HIP-graph replay: `ContactWorld.run(n, graph=True)` on chains of four links (joint Jacobian rebuilt
    on the device every step, a time-dependent push, with and without post-stabilisation) against the eager run, bit for bit."""
    from lcp_physics_amd import scenes
    B, n = 6, 25
    mass = torch.linspace(0.5, 1.5, B)
    a = scenes.make_chain_world(B, links=4, mass=mass, device=DEV, post_stab=post_stab)
    b = scenes.make_chain_world(B, links=4, mass=mass, device=DEV, post_stab=post_stab)
    for _ in range(n):
        a.step()
    b.run(n, graph=True)
    torch.cuda.synchronize()
    assert len(b._graphs) >= 1
    assert torch.equal(a.p, b.p) and torch.equal(a.v, b.v) and torch.equal(a.t, b.t) and torch.equal(a.Je, b.Je)
    assert torch.equal(a.contacts.count, b.contacts.count) and torch.equal(a.joints.jrot1, b.joints.jrot1)
    assert float(a.t.min()) > 0.4 and float((a.p[:, :4, 0].abs().max())) > 0.05          # the chain was hit and swings


def test_differentiable_rollout_captured_in_a_hip_graph_equals_the_eager_run():
    """A whole differentiable roll-out - `scenes.ChainWorlds.world()`, 12 `step(differentiable=True)` with joints and
    post-stabilisation, the loss and its backward - captured into one HIP graph (`torch.cuda.graph`): every launch of the C ABI
    goes to the capturing stream, nothing touches the host.  Replayed with new parameter values it must return the loss and the
    gradient of the eager run, bit for bit."""
    from lcp_physics_amd import scenes
    B, links, steps = 8, 4, 12
    chains = scenes.ChainWorlds(B, links=links, device=DEV)
    target = chains.p0 + torch.tensor([0.2, 20.0, -5.0], dtype=torch.float64, device=DEV)

    def loss_of(mass, push):
        world = chains.world(mass, push)
        for _ in range(steps):
            world.step(differentiable=True)
        return ((world.p - target) ** 2).mean(dim=(1, 2))

    mass = torch.linspace(0.5, 1.6, B, device=DEV).requires_grad_(True)
    push = (torch.tensor([0.0, 1.0, 0.05], device=DEV) * torch.linspace(0.8, 1.3, B, device=DEV).unsqueeze(1)).requires_grad_(True)
    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            mass.grad = push.grad = None
            loss_of(mass, push).sum().backward()
    torch.cuda.current_stream().wait_stream(side)
    mass.grad = push.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss_g = loss_of(mass, push)
        loss_g.sum().backward()
    with torch.no_grad():                                        # new parameter values in the captured tensors
        mass.mul_(1.1); push.mul_(0.95)
    g.replay()
    torch.cuda.synchronize()
    got = (loss_g.clone(), mass.grad.clone(), push.grad.clone())
    m2, p2 = mass.detach().clone().requires_grad_(True), push.detach().clone().requires_grad_(True)
    loss_e = loss_of(m2, p2)
    loss_e.sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(got[0], loss_e.detach()) and torch.equal(got[1], m2.grad) and torch.equal(got[2], p2.grad)
    assert float(m2.grad.abs().min()) > 0.0


def test_grad_demo_rollout_from_one_hip_graph_equals_the_eager_run_and_the_reference():
    """`ContactWorld.restart` + 36 differentiable steps + loss + backward of the batched `grad_demo` captured into ONE HIP graph:
    replayed, it returns the gradients of the eager run bit for bit - hence the reference's (rollout_grad.npz) to 1e-4."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rollout_grad.npz"))
    rep = 16
    world, force0 = _rollout_world(d, rep)
    p0, v0 = world.p.clone(), world.v.clone()
    a, b = [int(i) for i in d["loss_bodies"]]

    def loss_of():
        world.restart(p0, v0)
        for _ in range(int(d["nsteps"])):
            world.step(differentiable=True)
        pos = world.p[:, :, 1:]
        return (pos[:, a] - pos[:, b]).norm(dim=1)

    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            force0.grad = None
            loss_of().sum().backward()
    torch.cuda.current_stream().wait_stream(side)
    eager = force0.grad.clone()
    force0.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss_g = loss_of()
        loss_g.sum().backward()
    force0.grad.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(force0.grad, eager)
    gr = force0.grad.cpu().numpy()[::rep]
    err = np.abs(gr - d["grad"]).max(axis=1) / np.abs(d["grad"]).max(axis=1)
    assert err.max() <= 1e-4, err
    assert np.abs(loss_g.detach().cpu().numpy()[::rep] - d["loss"]).max() <= 1e-5 * np.abs(d["loss"]).max()


def test_joint_jacobian_backward_matches_autograd_of_the_torch_expression():
    """`lcp_joint_jacobian_backward_f64` (the backward of the joint Jacobian a differentiable step uses) against torch autograd through
    `JointSet.jacobian_torch`, the torch restatement of Joint.J() / FixedJoint.J() (constraints.py:26-50, 64-85): revolute joints with
    and without a second body, a fixed joint, one-row constraints and a pinned body in between (row bookkeeping)."""
    from lcp_physics_amd.physics.joints import JointSet
    B, nb = 33, 6
    g = torch.Generator().manual_seed(21)
    p0 = torch.zeros(nb, 3, dtype=torch.float64)
    p0[:, 1] = 40.0 * torch.arange(nb, dtype=torch.float64)
    p0[:, 2] = 25.0 * torch.rand(nb, generator=g, dtype=torch.float64)
    joints = [("total", 0), ("joint", 1, 0, (20.0, 3.0)), ("x", 2), ("joint", 2, None, (75.0, 11.0)), ("fixed", 3, 2), ("joint", 4, 3, (140.0, -6.0)),
              ("rot", 5), ("fixed", 5, None)]
    js = JointSet.from_list(joints, p0, B).to(DEV)
    p = (p0.unsqueeze(0) + 2.0 * torch.randn(B, nb, 3, generator=g, dtype=torch.float64)).to(DEV)
    rot = (js.jrot1 + 0.3 * torch.randn(B, js.jrot1.shape[1], generator=g, dtype=torch.float64).to(DEV))
    gJe = torch.randn(B, js.e, 3 * nb, generator=g, dtype=torch.float32).to(DEV)
    g_p, g_rot = js.jacobian_backward(nb, rot, gJe)
    pt, rt = p.clone().requires_grad_(True), rot.clone().requires_grad_(True)
    (js.jacobian_torch(pt, rt) * gJe.double()).sum().backward()
    torch.cuda.synchronize()
    assert float((g_p - pt.grad).abs().max()) <= 1e-12 * max(1.0, float(pt.grad.abs().max()))
    rev = js.revolute_mask
    assert float((g_rot - rt.grad * rev).abs().max()) <= 1e-12 * max(1.0, float(rt.grad.abs().max()))
    assert float((g_rot * (1 - rev)).abs().max()) == 0.0
    # the Jacobian the kernel computes at those angles is the one the expression describes
    js2 = JointSet(js.jtype, js.jb1, js.jb2, js.jr1, rot.clone(), js.e)
    assert float((js2.jacobian(p).double() - js.jacobian_torch(p, rot)).abs().max()) <= 1e-5


def test_state_update_backward_matches_autograd():
    """`lcp_state_update_backward_f64` (`_StateUpdate`: Body.move, the skipped vertex turn of a zero rotation increment, Joint.move -
    bodies.py:80-82, 199-202, constraints.py:39-43) against torch autograd of the three sums, both scales, with and without joints,
    missing cotangents."""
    from lcp_physics_amd.physics.batched_world import _StateUpdate
    from lcp_physics_amd.physics.joints import JointSet
    B, nb = 40, 5
    g = torch.Generator().manual_seed(8)
    p0 = torch.zeros(nb, 3, dtype=torch.float64)
    p0[:, 1] = 30.0 * torch.arange(nb, dtype=torch.float64)
    js = JointSet.from_list([("total", 0), ("joint", 1, 0, (10.0, 2.0)), ("joint", 3, 1, (50.0, 1.0)), ("fixed", 4, 3), ("joint", 1, 2, (40.0, 0.0))], p0, B).to(DEV)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    v = rn(B, nb, 3).float()
    v[:, 2, 0] = 0.0                                     # a body that does not turn: no geometry gradient through its angle
    v[::3, 4] = 0.0
    dt = (0.01 + 0.02 * torch.rand(B, generator=g, dtype=torch.float64)).to(DEV)
    for scale, with_joints, drop in ((1.0, True, None), (0.5, True, "g_g"), (1.0, False, None), (0.5, False, "g_p")):
        p, pg, rot = rn(B, nb, 3).to(DEV).requires_grad_(True), rn(B, nb, 3).to(DEV).requires_grad_(True), rn(B, js.jtype.shape[1]).to(DEV).requires_grad_(True)
        vv = v.to(DEV).clone().requires_grad_(True)
        p_out, rot_val = rn(B, nb, 3).to(DEV), rn(B, js.jtype.shape[1]).to(DEV)
        cot = [rn(B, nb, 3).to(DEV), rn(B, nb, 3).to(DEV), rn(B, js.jtype.shape[1]).to(DEV)]
        outs = _StateUpdate.apply(p, pg, rot if with_joints else None, vv, p_out, rot_val if with_joints else None, dt, scale, js if with_joints else None)
        assert torch.equal(outs[0], p_out) and torch.equal(outs[1], p_out) and (not with_joints or torch.equal(outs[2], rot_val))
        loss = 0
        if drop != "g_p":
            loss = loss + (outs[0] * cot[0]).sum()
        if drop != "g_g":
            loss = loss + (outs[1] * cot[1]).sum()
        if with_joints:
            loss = loss + (outs[2] * cot[2]).sum()
        loss.backward()
        got = [t.grad.clone() if t.grad is not None else None for t in (p, pg, rot, vv)]
        # the same three sums in torch
        p2, pg2, rot2, v2 = [t.detach().clone().requires_grad_(True) for t in (p, pg, rot, vv)]
        dp = v2.double() * scale * dt.view(-1, 1, 1)
        xy = (torch.arange(3, device=DEV) > 0).view(1, 1, 3)
        a = p2 + dp
        b = pg2 + torch.where((dp != 0) | xy, dp, dp.detach())
        ref = 0
        if drop != "g_p":
            ref = ref + (a * cot[0]).sum()
        if drop != "g_g":
            ref = ref + (b * cot[1]).sum()
        if with_joints:
            w1 = v2[:, :, 0].gather(1, js.jb1.long()).double() * scale * dt.view(-1, 1)
            ref = ref + ((rot2 + w1 * js.revolute_mask) * cot[2]).sum()
        ref.backward()
        torch.cuda.synchronize()
        assert float((got[3].double() - v2.grad.double()).abs().max()) <= 2e-6 * float(v2.grad.abs().max()), (scale, with_joints, drop)
        for k, t in ((0, p2), (1, pg2)):
            if t.grad is None:
                assert got[k] is None or float(got[k].abs().max()) == 0.0
            else:
                assert torch.equal(got[k], t.grad)
        if with_joints:
            assert torch.equal(got[2], rot2.grad)


@pytest.mark.parametrize("post_stab", [False, True])
def test_plain_steps_after_differentiable_steps_match_plain_steps(post_stab):
    """A differentiable step runs the kernels of a plain step (the state update is the kernels' own output, `_StateUpdate`), and the
    plain steps that follow it must not hand autograd-owned storage to a kernel as an output slot: 3 differentiable + 3 plain
    steps against 6 plain steps, bitwise, and the gradient of the third pose is still intact afterwards."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics.batched_world import ContactWorld
    B = 16
    w = scenes.make_drop_world(B, nbox=4, box=40.0, seed=3)
    geom = _geom([w["shapes"]] * B)
    g = lambda k: w[k].to(DEV)
    mk = lambda: ContactWorld(geom, g("p"), g("v"), g("Mdiag"), g("f"), g("rest"), g("fric"), Je=g("Je"), maxc=16, post_stab=post_stab)
    a, b = mk(), mk()
    b.f = b.f.clone().requires_grad_(True)
    for _ in range(6):
        a.step()
    for _ in range(3):
        b.step(differentiable=True)
    p3 = b.p
    p3_value = p3.detach().clone()
    (gref,) = torch.autograd.grad((p3 * p3).sum(), b.f, retain_graph=True)       # the gradient before anything else happens
    for _ in range(3):
        b.step()
    torch.cuda.synchronize()
    assert torch.equal(a.p, b.p.detach()) and torch.equal(a.v, b.v.detach()) and torch.equal(a.t, b.t)
    assert torch.equal(a.contacts.count, b.contacts.count)
    # the graph of the differentiable steps was not disturbed: neither the value of its output nor what its backward reads
    assert torch.equal(p3.detach(), p3_value)
    (g,) = torch.autograd.grad((p3 * p3).sum(), b.f)
    assert torch.equal(g, gref) and float(g.abs().max()) > 0

"""End-to-end throughput of the batched world WITH contact detection: B scenes of a floor
and 4 boxes settling into a stack, `ContactWorld.step()` = lcp_solve_dynamics_f32 + lcp_move_find_contacts_f64.
Prints one JSON line (simulation steps / s, the two kernels' times from HIP events, the CPU oracle on a sample).

    python tools/bench_world.py [--batch 4096] [--steps 100] [--settle 40]
"""
import argparse, json, os, sys, time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--nbox", type=int, default=4)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--settle", type=int, default=40, help="untimed steps before the timed region (contacts form)")
    ap.add_argument("--cpu-scenes", type=int, default=2)
    ap.add_argument("--maxc", type=int, default=16, help="contact capacity per scene")
    ap.add_argument("--box", type=float, default=40.0)
    ap.add_argument("--graph", action="store_true", help="time ContactWorld.run(steps, graph=True): HIP graph replay")
    ap.add_argument("--post-stab", action="store_true", help="with post-stabilisation (world.py:109-121; off by default as in the reference)")
    ap.add_argument("--record", type=int, default=0, metavar="K",
                    help="also time RECORDED roll-outs: K steps of ContactWorld.step(differentiable=True) from the settled state + loss.backward(), "
                         "under torch.cuda.set_sync_debug_mode('error') (a synchronising torch call in a step or in the backward fails the run)")
    ap.add_argument("--record-reps", type=int, default=5)
    ap.add_argument("--no-strict", action="store_true", help="World(strict_no_pen=False): the retry loop of world.py:88-101 stops at dt / 4")
    args = ap.parse_args()
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics import batched_world as bw
    from lcp_physics_amd.physics import contacts as ct
    dev = torch.device("cuda")
    w = scenes.make_drop_world(args.batch, nbox=args.nbox, box=args.box)
    geom = ct.GeometryBatch.from_shapes(w["shapes"], args.batch).to(dev)
    g = lambda k: w[k].to(dev)
    world = bw.ContactWorld(geom, g("p"), g("v"), g("Mdiag"), g("f"), g("rest"), g("fric"), Je=g("Je"), maxc=args.maxc, post_stab=args.post_stab,
                            strict_no_penetration=not args.no_strict)
    for _ in range(args.settle):
        world.step()
    world.check_capacity()
    torch.cuda.synchronize()
    counts0 = world.contacts.count.float()
    # split timing: events around each launch
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    solve, move = bw.solve_dynamics, ct.move_and_find_contacts
    k_ = {"i": 0}

    def solve_t(*a, **kw):
        ev[k_["i"]][0].record(); out = solve(*a, **kw); ev[k_["i"]][1].record(); return out

    def move_t(*a, **kw):
        out = move(*a, **kw); ev[k_["i"]][2].record(); return out

    bw.solve_dynamics, ct.move_and_find_contacts = solve_t, move_t
    if args.graph:
        bw.solve_dynamics, ct.move_and_find_contacts = solve, move
        world.run(4, graph=True)                                         # capture outside the timed region
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        world.run(args.steps, graph=True)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        for e in ev:
            for x in e:
                x.record()
        torch.cuda.synchronize()
    else:
        t0 = time.perf_counter()
        for i in range(args.steps):
            k_["i"] = i
            world.step()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    bw.solve_dynamics, ct.move_and_find_contacts = solve, move
    world.check_capacity()
    # (per-launch events cannot be timed inside a replayed graph: the split is reported for eager runs only)
    solve_ms = None if args.graph else sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    move_ms = None if args.graph else sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    out = {"metric": "sim steps/s with contact detection (ContactWorld.step)", "value": args.batch * args.steps / wall,
           "unit": "sim steps/s", "batch": args.batch, "steps": args.steps, "hip_graph": bool(args.graph), "post_stab": bool(args.post_stab),
           "strict_no_penetration": not args.no_strict, "ms_per_step": wall / args.steps * 1e3,
           "solve_dynamics_ms": solve_ms, "move_find_contacts_ms": move_ms,
           "mean_contacts_start": float(counts0.mean()), "mean_contacts_end": float(world.contacts.count.float().mean()),
           "max_contacts": int(world.contacts.count.max()), "mean_trials_last_step": float(world.contacts.trials.float().mean()),
           "mean_t": float(world.t.mean()), "nonzero_status": int((world._out["status"] != 0).sum())}
    if args.record > 0:
        # recorded roll-outs from the state the timed steps ended in: every step an autograd node (SolveDynamicsFunction, ContactFrameFunction,
        # _StateUpdate), one loss on the final pose, one backward through all of them
        import warnings
        p_end, v_end = world.p.detach().clone(), world.v.detach().clone()
        times = []
        mode = torch.cuda.get_sync_debug_mode()
        gn = None
        for rep in range(args.record_reps + 1):                          # (the first repetition is a warm-up)
            Md = g("Mdiag").requires_grad_(True)
            v0 = v_end.clone().requires_grad_(True)
            p0 = p_end.clone().requires_grad_(True)
            rw = bw.ContactWorld(geom, p0, v0, Md, g("f"), g("rest"), g("fric"), Je=g("Je"), maxc=args.maxc, post_stab=args.post_stab,
                                 strict_no_penetration=not args.no_strict, check=False)
            torch.cuda.synchronize()
            with warnings.catch_warnings():
                warnings.simplefilter("error", RuntimeWarning)           # (the dense boundary announces itself with one)
                torch.cuda.set_sync_debug_mode("error")
                try:
                    t0 = time.perf_counter()
                    for _ in range(args.record):
                        rw.step(differentiable=True)
                    loss = (rw.p[:, 1:, 1:] ** 2).sum() * 1e-4
                    loss.backward()
                finally:
                    torch.cuda.set_sync_debug_mode(mode)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            gn = [float(t.abs().max()) for t in (Md.grad, v0.grad, p0.grad)]
            assert all(x == x and x < float("inf") for x in gn), gn
        best = min(times[1:])
        out["recorded"] = {"steps_per_rollout": args.record, "reps": args.record_reps, "value": args.batch * args.record / best,
                           "unit": "recorded sim steps/s (K differentiable steps + one backward through them)",
                           "ms_per_rollout": best * 1e3, "ms_per_rollout_all": [t * 1e3 for t in times[1:]],
                           "host_synchronisations": 0, "sync_debug_mode": "error", "runtime_warnings": 0,
                           "max_abs_grad_Mdiag_v0_p0": gn,
                           "has_fused_backward": bool(bw._lib.load().lcp_step_has_backward(world.nb, args.maxc, world.e, bw._lib.COMPUTE_F64))}
    # CPU oracle on a few of the same scenes (same number of steps from the same start)
    if args.cpu_scenes > 0:
        from oracle import contacts_oracle as C
        from oracle import world_oracle as W
        t1 = time.perf_counter()
        nsteps = min(args.settle + args.steps, 30)
        for s in range(args.cpu_scenes):
            d = lambda k: w[k][s].double().numpy()
            p, v = d("p"), d("v")
            cs = C.find_contacts(W.bodies_at(w["shapes"], p), eps=0.1)
            for _ in range(nsteps):
                p, v, cs, _, _ = W.step_dt(w["shapes"], p, v, cs, d("Mdiag"), d("f"), d("rest"), d("fric"), d("Je"), world.dt)
        cpu = time.perf_counter() - t1
        out["cpu_oracle"] = {"value": args.cpu_scenes * nsteps / cpu, "unit": "sim steps/s", "cores": 1,
                             "sample": "%d scenes x %d steps, oracle/world_oracle.py (numpy + torch fp64)" % (args.cpu_scenes, nsteps)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

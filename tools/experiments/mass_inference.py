"""The reference's `experiments/inference.py:26-89` on a batch: recover the mass of a chain's links from an observed trajectory by
gradient descent through the simulator.  B scenes start from different guesses; every iteration rolls all of them out
(`ContactWorld.step(differentiable=True)`: solve_dynamics, joints, contacts and post-stabilisation in the graph) and
back-propagates the mean squared distance to the observed poses (RMSprop, lr 0.5, as in the reference, on log-mass).

    python tools/experiments/mass_inference.py [--batch 64] [--links 10] [--steps 36] [--iters 40] [--graph]

`--graph` captures one whole iteration (all steps forward and backward) into a HIP graph and replays it: the kernels and the
results are the same, the host issues one launch per iteration instead of about 9 000.

Prints one JSON line: the recovered masses, the wall time per iteration (forward + backward of B roll-outs)."""
import json, os, sys, time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from lcp_physics_amd import scenes
    a = sys.argv[1:]
    opt = lambda k, d: type(d)(a[a.index(k) + 1]) if k in a else d
    B, links, steps, iters, true_mass = opt("--batch", 64), opt("--links", 10), opt("--steps", 36), opt("--iters", 40), opt("--mass", 0.7)
    dev = "cuda"

    graph = "--graph" in a
    chains = {}

    def rollout(mass):
        n = mass.shape[0]
        if n not in chains:
            chains[n] = scenes.ChainWorlds(n, links=links, device=dev)               # (host work once; `world()` is device work only)
        world = chains[n].world(mass)
        poses = []
        for _ in range(steps):
            world.step(differentiable=True)
            poses.append(world.p)
        return torch.stack(poses, 1), world

    with torch.no_grad():
        observed, _ = rollout(torch.full((1,), true_mass, device=dev))
    g = torch.Generator().manual_seed(0)
    log_m = torch.log(0.3 + 1.7 * torch.rand(B, generator=g)).to(dev).requires_grad_(True)
    start = log_m.detach().exp().cpu()
    optim = torch.optim.RMSprop([log_m], lr=0.05)
    times, hist = [], []
    g = None
    if graph:
        # one HIP graph for the whole iteration - 36 differentiable steps forward and backward, about 9 000 launches
        rollout(log_m.detach().exp())                                                    # (builds the prototype outside the capture)
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)        # (log_m was created on the default stream)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                                                              # warm-up on a side stream
                optim.zero_grad(set_to_none=True)
                poses, world = rollout(log_m.exp())
                ((poses - observed) ** 2).mean(dim=(1, 2, 3)).sum().backward()
        torch.cuda.current_stream().wait_stream(side)
        optim.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            poses, world = rollout(log_m.exp())
            loss = ((poses - observed) ** 2).mean(dim=(1, 2, 3))
            loss.sum().backward()
    for it in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if g is not None:
            g.replay()
            optim.step()
        else:
            optim.zero_grad()
            poses, world = rollout(log_m.exp())
            loss = ((poses - observed) ** 2).mean(dim=(1, 2, 3))
            loss.sum().backward()
            optim.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        hist.append(float(loss.detach().mean()))
        if os.environ.get("LCP_VERBOSE"):
            print(it, hist[-1], log_m.detach().exp()[:4].cpu().tolist(), log_m.grad[:4].cpu().tolist())
    launches = None
    if "--count" in a and g is None:                  # device launches (kernels, copies, memsets) of one eager iteration, per simulated step
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            optim.zero_grad()
            poses, world = rollout(log_m.exp())
            ((poses - observed) ** 2).mean(dim=(1, 2, 3)).sum().backward()
            torch.cuda.synchronize()
        ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        launches = len(ev) / steps
        if "--names" in a:
            names = {}
            for e in ev:
                names[e.name[:90]] = names.get(e.name[:90], 0) + 1
            for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:40]:
                print("%7.1f  %s" % (v / steps, k), file=sys.stderr)
    m = log_m.detach().exp().cpu()
    times = sorted(times[2:])
    print(json.dumps({"experiment": "mass inference through the simulator (experiments/inference.py), batched", "batch": B, "links": links,
                      "equality_rows": 2 * links, "steps_per_rollout": steps, "iterations": iters, "true_mass": true_mass,
                      "start_mass_min_max": [float(start.min()), float(start.max())],
                      "recovered_mass_median": float(m.median()), "recovered_within_2pct": float(((m - true_mass).abs() < 0.02 * true_mass).float().mean()),
                      "loss_first_last": [hist[0], hist[-1]], "s_per_iteration_median": times[len(times) // 2],
                      "device_launches_per_step_fwd_bwd_eager": launches,
                      "sim_steps_fwd_bwd_per_s": B * steps / times[len(times) // 2], "status_flags": int(world.sticky_status.max()), "hip_graph": bool(graph)}))


if __name__ == "__main__":
    main()

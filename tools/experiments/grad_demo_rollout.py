"""Time of one optimisation iteration of the batched `grad_demo` (demos/grad_demo.py:19-83: three balls, 36 steps, loss = distance to
the target, gradient with respect to the initial push) - eager, and replayed from ONE HIP graph (`ContactWorld.restart` + the
steps + the loss + its backward captured with `torch.cuda.graph`).  Scenes: tests/golden/rollout_grad.npz replicated.

    python tools/experiments/grad_demo_rollout.py [--rep 128]                       # B = 8 x rep scenes, replay of the captured graph
    python tools/experiments/grad_demo_rollout.py [--rep 128] --eager [--count]     # eager timing (and device launches per step)
"""
import json, os, sys, time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from tests.test_hip_contacts import _rollout_world
    a = sys.argv[1:]
    rep = int(a[a.index("--rep") + 1]) if "--rep" in a else 128
    d = np.load(os.path.join(ROOT, "tests", "golden", "rollout_grad.npz"))
    world, force0 = _rollout_world(d, rep)
    p0, v0 = world.p.clone(), world.v.clone()
    i, j = [int(k) for k in d["loss_bodies"]]
    nsteps = int(d["nsteps"])

    def loss_of():
        world.restart(p0, v0)
        for _ in range(nsteps):
            world.step(differentiable=True)
        pos = world.p[:, :, 1:]
        return (pos[:, i] - pos[:, j]).norm(dim=1)

    def timed(fn, n=8):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    def eager():
        force0.grad = None
        loss_of().sum().backward()

    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    B = force0.shape[0]
    rel = lambda: float((np.abs(force0.grad.cpu().numpy()[::rep] - d["grad"]).max(axis=1) / np.abs(d["grad"]).max(axis=1)).max())
    if "--eager" in a:
        # eager roll-outs in a process of their own (no side stream, no capture: mixing them with the default stream's allocations in one
        # process made the eager timing depend on the order of the two)
        eager(); eager()
        t_eager = timed(eager)
        out = {"experiment": "batched grad_demo: %d steps forward + backward, eager" % nsteps, "batch": B, "s_per_iteration_eager": t_eager,
               "sim_steps_fwd_bwd_per_s_eager": B * nsteps / t_eager, "worst_relative_gradient_error_vs_reference_autograd": rel()}
        if "--count" in a:         # device launches (kernels + copies + memsets) of one eager iteration, per simulated step
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
                eager(); torch.cuda.synchronize()
            ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
            out["device_launches_per_step_fwd_bwd_eager"] = len(ev) / nsteps
            if "--names" in a:
                names = {}
                for e in ev:
                    names[e.name[:60]] = names.get(e.name[:60], 0) + 1
                for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:25]:
                    print("%6d  %s" % (v, k), file=sys.stderr)
        print(json.dumps(out))
        return
    side = torch.cuda.Stream()                                  # (warm-up of the capture on a side stream, as torch asks for)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager(); eager()
    torch.cuda.current_stream().wait_stream(side)
    ref = force0.grad.clone()
    force0.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss_of().sum().backward()
    t_graph = timed(g.replay)
    print(json.dumps({"experiment": "batched grad_demo: %d steps forward + backward, one HIP graph" % nsteps, "batch": B,
                      "s_per_iteration_hip_graph": t_graph, "sim_steps_fwd_bwd_per_s_hip_graph": B * nsteps / t_graph,
                      "graph_equals_eager_bitwise": bool(torch.equal(force0.grad, ref)),
                      "worst_relative_gradient_error_vs_reference_autograd": rel()}))


if __name__ == "__main__":
    main()

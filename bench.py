#!/usr/bin/env python
"""bench.py - the BASELINE.json metric on MI355X.

metric : sim steps/sec at batch=4096 x 16 contacts, forward + backward (implicit diff)
step   : one pass of the hot path over one batch of synthetic scenes already resident in HBM
         (one sim step = assembly + one LCP solve + integrate; fwd+bwd adds the
         implicit-differentiation backward): the fused step kernel lcp_step_fused_f32 (contact list ->
         new velocities and poses) + lcp_pdipm_backward_f32, for B = 4096 scenes per GPU (floor + 4-box
         stack, 4 contact points per interface: nz 15, nineq 64, neq 3).  `--mode dense` times the dense
         LCPFunction boundary instead (lcp_pdipm_forward_f32 on pre-assembled (Q,p,G,h,A,b,F) + backward).
N GPUs : one process per GPU (torchrun), every rank owns its own 4096 scenes (weak scaling,
         config 4 = 8 x 4096); no collective on the solve path - torch.distributed (RCCL) is only
         used for the barriers and the MAX-over-ranks wall time.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - dominant kernel (the forward PDIPM kernel): algorithmic FLOPs (SURVEY.md §8d,
                 lcp_physics_amd/flops.py, with the iteration counts the kernel reports) divided by
                 its average launch duration measured with HIP events on the launch stream inside
                 the timed region; peak = the FP64 vector/matrix rate when the parity path
                 (fp64 arithmetic) runs, the FP32 rate for --compute f32.
  cpu_baseline - the oracle (a port, torch CPU fp64) timed on this host's cores on the same
                 workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}   # MI355X FP64 / FP32 vector = matrix rates (datasheet; MI355X_MICROARCH.md lists FP32)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="scenes per GPU")
    ap.add_argument("--nbox", type=int, default=4)
    ap.add_argument("--pts", type=int, default=4)
    ap.add_argument("--compute", default="f64", choices=["f64", "f32"])
    ap.add_argument("--mode", default="fused", choices=["dense", "fused"])
    ap.add_argument("--bwd", default="dense", choices=["dense", "physical"],
                    help="backward timed in the step: 'dense' = LCPFunction.backward (7 dense gradients, lcp.py:37-64); "
                         "'physical' (fused mode only) = lcp_step_backward_f32, gradients w.r.t. the physical inputs")
    ap.add_argument("--fwd-only", action="store_true",
                    help="time the forward only (BASELINE configs[1] is forward-only); the default is the headline fwd+bwd")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU work for the baseline")
    return ap.parse_args()


def cpu_baseline(sc_cpu, cot, budget_s=15.0, fwd_only=False):
    """Oracle (port of the reference algorithm, vectorised torch fp64) on the host cores: forward +
    backward on a bounded sample of the same scenes (sized from a calibration pass to ~budget_s)."""
    from oracle import pdipm_oracle as O
    threads = max(1, min(os.cpu_count() or 1, 16))      # tiny batched ops: more threads only add overhead
    torch.set_num_threads(threads)

    def run(n):
        sub = sc_cpu.slice(0, n).to(dtype=torch.float64)
        lcp = O.assemble_lcp(*sub.assembly_args())
        t0 = time.perf_counter()
        sol = O.lcp_forward(*lcp)
        if not fwd_only:
            O.lcp_backward(sol, *lcp, cot[:n].double())
        return time.perf_counter() - t0

    run(32)                                             # warm-up
    cal = run(128)
    sample = int(max(128, min(sc_cpu.B, 128 * budget_s / max(cal, 1e-3))))
    total, passes = 0.0, 0
    while total < budget_s * 0.66 and passes < 64:      # repeat passes over the sample up to ~10 s of CPU work
        total += run(sample)
        passes += 1
    return {"value": sample * passes / total, "unit": "sim steps/s", "cores": threads, "kind": "port",
            "sample": "%d of the same scenes x %d passes, %s, vectorised torch fp64 oracle, %.2f s"
                      % (sample, passes, "forward only" if fwd_only else "fwd+bwd", total)}


def parity_vs_oracle(sc_cpu, cot, x_gpu, z_gpu, s_gpu, dp_gpu, n=256):
    """The metric's second half ("fwd+bwd rel-err vs ref"): the step just timed against the fp64 oracle on the first
    `n` scenes (identical fp32 inputs).  err_x = |x - x_ref| / max(|x_ref|, |Q^-1 p|) per scene; the
    backward error is taken on dp = dx (lcp.py:52), scaled by |Q^-1 dl_dx|, over the scenes where the oracle's own
    backward system is well posed (same mask as tests/test_hip_parity.py::test_stack_scenes_backward_parity)."""
    from oracle import pdipm_oracle as O
    from tests import parity
    n = min(n, sc_cpu.B)
    # identical LCP inputs: assembled in fp32 exactly as the step kernel assembles them (engines.py:50-74 in the
    # data's dtype; tests/test_hip_parity.py::test_assembly_kernel_matches_oracle), then solved in fp64
    lcp = [None if t is None else t.double() for t in O.assemble_lcp(*sc_cpu.slice(0, n).assembly_args())]
    Q, p, G, h, A, b, F = lcp
    ref = O.lcp_forward(*lcp)
    ex = parity.err_x(x_gpu[:n].double().cpu(), ref.x, Q, p)
    out = {"scenes": n, "tolerance": 1e-4, "fwd_err_x_max": float(ex.max()), "fwd_err_x_median": float(ex.median())}
    # contact index sets {i : z_i > s_i}: compared wherever the oracle's own decision is not a tie between
    # two numbers that both converged to zero (tests/test_hip_parity.py::_decisive)
    z, sl = z_gpu[:n].double().cpu(), s_gpu[:n].double().cpu()
    big = torch.maximum(ref.z.abs(), ref.s.abs())
    nondeg = torch.maximum(ref.z.abs() / ref.z.abs().max(dim=1, keepdim=True)[0],
                           ref.s.abs() / ref.s.abs().max(dim=1, keepdim=True)[0]) > 1e-5
    dec = ((ref.z - ref.s).abs() > 1e-3 * big) & nondeg
    out["index_set_mismatches"] = int((((z > sl) != (ref.z > ref.s)) & dec).sum())
    out["index_set_rows_compared"] = int(dec.sum())
    if dp_gpu is not None:
        c64 = cot[:n].double()
        gref = O.lcp_backward(ref, *lcp, c64)
        res = parity.kkt_backward_residual(Q, G, A, F, ref.z, ref.s, c64, gref["dp"], -gref["dh"], -gref["db"])
        ok = torch.stack([v for v in res.values()]).max(dim=0)[0] < 1e-9
        zs, ss = ref.z.max(dim=1, keepdim=True)[0], ref.s.max(dim=1, keepdim=True)[0]
        ok = ok & (torch.maximum(ref.z / zs, ref.s / ss).min(dim=1)[0] > 1e-6)
        fl = parity.grad_floors(Q, p, c64, ref.x, ref.z, ref.y)
        eg = parity.err_grads({"p": dp_gpu[:n].double().cpu()}, {"p": gref["dp"]}, fl)["p"]
        if bool(ok.any()):
            out.update({"bwd_err_dp_max": float(eg[ok].max()), "bwd_err_dp_median": float(eg[ok].median())})
        out["bwd_well_posed_scenes"] = int(ok.sum())
    return out


def main():
    args = parse()
    from lcp_physics_amd import flops, scenes, shard
    from lcp_physics_amd.lcp import lcp_backward, lcp_solve
    from lcp_physics_amd.physics import assemble_contacts, fused_step
    from lcp_physics_amd.physics.batched_world import fused_step_backward, solution_of_step

    rank, local_rank, world = shard.init_process_group()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    B = args.batch
    nb, nc = args.nbox + 1, args.nbox * args.pts
    nz, m, e = 3 * nb, 4 * nc, 3

    # synthetic scenes, generated on the host, then resident in HBM before any timing
    sc_cpu = scenes.make_stack_scenes(B=B, nbox=args.nbox, pts_per_interface=args.pts, seed=1236 + 1000 * rank,
                                      dtype=torch.float32)
    sc = sc_cpu.to(device=dev)
    g = torch.Generator().manual_seed(4321 + rank)
    cot_cpu = torch.randn(B, nz, generator=g, dtype=torch.float32)
    cot = cot_cpu.to(dev)
    lcp = assemble_contacts(sc)            # dense (Q,p,G,h,A,b,F) in HBM, built by the HIP assembly kernel
    G, A = lcp[2], lcp[4]
    sol = lcp_solve(*lcp, compute=args.compute)
    grads = lcp_backward(sol, cot)
    step_out = fused_step(sc, compute=args.compute) if args.mode == "fused" else None
    torch.cuda.synchronize()

    # (the output buffers and the workspace are re-used every step, so the handle the backward takes is built once)
    step_sol = solution_of_step(sc, step_out, G, A, compute=args.compute) if args.mode == "fused" else None

    if args.bwd == "physical" and args.mode != "fused":
        raise SystemExit("--bwd physical needs --mode fused")
    cot_v = (-cot).reshape(B, nb, 3).contiguous()          # d(loss)/d(v_new) = -d(loss)/dx
    pgrads = None

    def one_step(ev=None):
        nonlocal sol, step_out
        if ev is not None:
            ev[0].record()
        if args.mode == "dense":
            lcp_solve(*lcp, compute=args.compute, ws=sol.ws, out=sol)
            s_ = sol
        else:
            step_out = fused_step(sc, compute=args.compute, ws=step_out["ws"], out=step_out)
            s_ = step_sol
        if ev is not None:
            ev[1].record()
        if args.fwd_only:
            pass
        elif args.bwd == "physical":
            nonlocal pgrads
            pgrads = fused_step_backward(sc, step_out, cot_v, compute=args.compute, grads=pgrads)
        else:
            lcp_backward(s_, cot, out=grads)
        if ev is not None:
            ev[2].record()

    for _ in range(args.warmup):
        one_step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    torch.cuda.synchronize()
    shard.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(events[k])
    torch.cuda.synchronize()
    shard.barrier()
    wall = time.perf_counter() - t0
    wall = shard.max_over_ranks(wall, device=dev)

    fwd_ms = sum(ev[0].elapsed_time(ev[1]) for ev in events) / args.steps
    bwd_ms = sum(ev[1].elapsed_time(ev[2]) for ev in events) / args.steps
    iters = (sol.iters if args.mode == "dense" else step_out["iters"]).double()
    status = (sol.status if args.mode == "dense" else step_out["status"])
    mean_it = float(iters.mean())
    fl_fwd = float(sum(flops.flops_forward(nz, m, e, it) for it in iters.cpu().tolist()))
    fl_bwd = flops.flops_backward(nz, m, e) * B
    total_steps = B * world * args.steps
    value = total_steps / wall
    achieved = fl_fwd / (fwd_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.compute]
    alg_bytes = (flops.bytes_fused_step(nb, nc) if args.mode == "fused" else flops.bytes_forward(nz, m, e)) * B
    # HBM traffic of the forward kernel: measured separately with rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE in their
    # own passes; FETCH_SIZE doubled per MI355X_MICROARCH.md) and committed under profiles/; bench.py cannot run
    # the counters itself, so it quotes that file when it matches this configuration.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath)).get("%s_B%d_nc%d_%s" % (args.mode, B, nc, args.compute))
        if tj:
            traffic = (2 * tj["fetch_kb"] + tj["write_kb"]) * 1024.0
    # which BASELINE.json config the flags amount to (the default run is configs[2], the one the metric is quoted on)
    cfg = {(1024, 8): "configs[1]", (4096, 16): "configs[2]", (32768, 16): "configs[3] on one GPU"}.get(
        (B, nc), "configs[3]" if (B * world, nc) == (32768, 16) else "variant")
    what = "forward only" if args.fwd_only else "forward + backward (implicit diff)"
    out = {
        "metric": "sim steps/sec at batch=%dx%d contacts, %s" % (B, nc, "fwd" if args.fwd_only else "fwd+bwd"),
        "value": value,
        "unit": "sim steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.compute,
        "data": "synthetic",
        "config": {"workload": "%s: batch=%d x %d contacts (%d-box stack, %d pts/interface; nz %d, nineq %d, "
                               "neq %d) per GPU, fp32 I/O, LCP %s, mode=%s, bwd=%s"
                               % (cfg, B, nc, args.nbox, args.pts, nz, m, e, what, args.mode,
                                  "none" if args.fwd_only else args.bwd),
                   "global_batch": B * world, "parallelism": "scenes sharded x%d, no collectives" % world,
                   "mean_pdipm_iters": mean_it, "nonzero_status": int((status != 0).sum())},
        "roofline": {"bound": "mfma",
                     "kernel": "lcp_fwd_quad<float,%s,%s> (PDIPM forward%s)" % (
                         "double" if args.compute == "f64" else "float", "true" if args.mode == "fused" else "false",
                         ", fused assembly + integrate" if args.mode == "fused" else "; the event-timed forward call also contains the classify launch"),
                     "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                     "fwd_ms": fwd_ms, "bwd_ms": bwd_ms,
                     "bwd_achieved": fl_bwd / (bwd_ms * 1e-3) / 1e12,
                     "algorithmic_flops_per_launch": fl_fwd,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "hbm_frac_algorithmic": alg_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sc_cpu, cot_cpu, args.cpu_budget, args.fwd_only)
        x_gpu = sol.x if args.mode == "dense" else -step_out["v_new"].reshape(B, nz)
        dp_gpu = None if (args.fwd_only or args.bwd == "physical") else grads[1]
        zs_src = sol if args.mode == "dense" else None
        z_gpu = zs_src.z if zs_src is not None else step_out["z"]
        s_gpu = zs_src.s if zs_src is not None else step_out["s"]
        out["parity"] = parity_vs_oracle(sc_cpu, cot_cpu, x_gpu, z_gpu, s_gpu, dp_gpu)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

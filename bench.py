#!/usr/bin/env python
"""bench.py - the BASELINE.json metric on MI355X.

metric : sim steps/sec at batch=4096 x 16 contacts, forward + backward (implicit diff)
step   : one pass of the hot path over one batch of synthetic scenes already resident in HBM
         (one sim step = assembly + one LCP solve + integrate; fwd+bwd adds the
         implicit-differentiation backward): the fused step kernel lcp_step_fused_f32 (contact list ->
         new velocities and poses) + lcp_pdipm_backward_f32, for B = 4096 scenes per GPU (floor + 4-box
         stack, 4 contact points per interface: nz 15, nineq 64, neq 3).  `--mode dense` times the dense
         LCPFunction boundary instead (lcp_pdipm_forward_f32 on pre-assembled (Q,p,G,h,A,b,F) + backward).
N GPUs : one process per GPU, every rank owns its own 4096 scenes (weak scaling, config 4 = 8 x 4096); no
         collective on the solve path - torch.distributed (RCCL) is only used for the barriers and the
         MAX-over-ranks wall time.  `python bench.py --gpus N` started WITHOUT a launcher starts the N ranks
         itself (torch.distributed.run on 127.0.0.1, one per visible device) and refuses when fewer than N
         devices are visible; under torchrun (the driver's form) --gpus must equal WORLD_SIZE.  `n_gpus` in
         the output is the number of ranks that reported, on distinct devices.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      - dominant kernel (the forward PDIPM kernel).  `achieved` / `frac` = the FLOPs the kernel really EXECUTES
                  (lcp_physics_amd/flops.py: body-space system of the pinned variant - formation + LU of nz - neq rows per
                  iteration, two KKT solves, residuals - with the iteration counts the kernel reports) divided by its average
                  launch duration, measured with HIP events on the launch stream inside the timed region, over the FP64 VECTOR
                  rate (the kernel issues no MFMA: `bound` says "valu_fp64").  `frac_algorithmic` = SURVEY.md §8d's count of
                  the REFERENCE's dense formulation over the same time and peak: it exceeds 1 because the kernel factors
                  nz - neq = 12 rows where the reference factors nineq = 64 - a different algorithm with the same answers,
                  not an efficiency.  `insts_per_useful_fma` cross-checks the executed model against the SQ_INSTS_VALU counter;
                  `traffic` = HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE counters; counters and registers are
                  quoted from the committed rocprofv3 / compiler reports under profiles/ (bench.py cannot run --pmc itself).
                  `roofline.bwd` = the backward kernel against the HBM roofline (it is bound by the 21.6 KB of dense
                  gradients it writes per scene): `achieved` = measured traffic (else algorithmic bytes) / its event time.
  cpu_baseline  - the oracle (a port, torch CPU fp64) timed on this host's cores on the same workload (rank 0,
                  N=1 only);  cpu_reference - the UNMODIFIED reference timed in the build container
                  (profiles/r01_reference_cpu_timing.json; /root/reference does not exist on the GPU box).
  sustained     - the same step repeated for >= 1 s after the timed region (independent evidence of GPU work).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EVENT_STRIDE = 8
PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}   # MI355X FP64 / FP32 vector = matrix rates (datasheet; MI355X_MICROARCH.md lists FP32)
HBM_PEAK_GBS = 8000.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: enough warm-up for the clocks to settle (the first tens of launches of a cold device run ~10 % slower) and a
    # timed region of ~40 ms; the whole default run still takes seconds
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
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
    ap.add_argument("--sustain", type=float, default=1.0, help="seconds of the untimed sustained loop after the timed steps (0 = off)")
    ap.add_argument("--share-devices", action="store_true",
                    help="TESTING AID: let several ranks use one GPU (gloo instead of RCCL); the line then says devices_used < n_gpus")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ rank protocol
def timed_steps(work, steps, warmup, sync, reduce_dev):
    """The contract's timing rule: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier +
    device synchronisation on both sides; returns the MAX wall time over the ranks and the per-step events."""
    from lcp_physics_amd import shard
    for _ in range(warmup):
        work.step()
    # HIP events bracket the kernels of every EVENT_STRIDE-th timed step only: an event record is a packet of its own in the queue
    # and costs the stream 2-3 us - three of them on every step were 4 % of the headline step and 15 % of a configs[1] step
    # (never the first timed step: the queue is empty after the barrier, and the gap between its event record and the arrival of its
    #  first launch would be booked as kernel time - 0.089 against 0.068 ms measured on the headline)
    sampled = set(range(EVENT_STRIDE - 1, steps, EVENT_STRIDE)) or {steps - 1}
    events = [work.new_events() if k in sampled else None for k in range(steps)]
    sync()
    shard.barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        work.step(events[k])
    sync()
    own = time.perf_counter() - t0                   # this rank's K steps, device work included, before it waits for the others
    shard.barrier()
    wall = time.perf_counter() - t0
    return shard.max_over_ranks(wall, device=reduce_dev), events, own


def run_rank(args, make_work, device=None):
    """Everything one rank does.  `make_work(args, rank, device)` builds the workload (HipStackWorkload below; the CPU
    test of the N > 1 protocol passes a stand-in whose step launches nothing).  Returns the JSON object on rank 0."""
    from lcp_physics_amd import shard
    rank, local_rank, world = shard.env_rank()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    shard.init_process_group(backend="gloo" if (args.share_devices or device is not None) else None)
    dev = device if device is not None else shard.rank_device(local_rank, share_devices=args.share_devices)
    is_gpu = torch.device(dev).type == "cuda"
    if is_gpu:
        torch.cuda.set_device(dev)
    rdev = shard.reduce_device(dev)
    sync = torch.cuda.synchronize if is_gpu else (lambda: None)
    work = make_work(args, rank, dev)
    wall, events, own_wall = timed_steps(work, args.steps, args.warmup, sync, rdev)
    # how many ranks really ran, and on how many distinct devices
    reported = int(round(shard.sum_over_ranks(1.0, device=rdev)))
    dev_index = torch.device(dev).index or 0
    devices = shard.gather_scenes(torch.tensor([[float(dev_index)]], dtype=torch.float64, device=rdev), world, device=rdev) \
        if world > 1 else torch.tensor([[float(dev_index)]])
    devices_used = len(set(int(v) for v in devices.flatten().tolist()))
    if reported != world:
        raise SystemExit("bench.py: %d of %d ranks reported" % (reported, world))
    if is_gpu and devices_used != world and not args.share_devices:
        raise SystemExit("bench.py: %d ranks on %d distinct devices" % (world, devices_used))
    sustained = None
    if args.sustain > 0:
        sync()
        t0, n = time.perf_counter(), 0
        while True:
            for _ in range(max(1, args.steps)):
                work.step()
            n += max(1, args.steps)
            sync()
            if time.perf_counter() - t0 >= args.sustain:
                break
        dt = time.perf_counter() - t0
        sustained = {"seconds": dt, "steps": n, "value": work.units_per_step * world * n / dt, "unit": "sim steps/s",
                     "note": "untimed-region repeat of the same step (this rank's clock), not the metric"}
    total_units = work.units_per_step * world * args.steps
    out = {
        "metric": work.metric_name(),
        "value": total_units / wall,
        "unit": "sim steps/s",
        "n_gpus": reported,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.compute,
        "data": "synthetic",
    }
    rep_ = work.report(events, world)                 # (every rank: its own kernel timings go into `per_rank`)
    out.update(rep_)
    # per-rank record, so that an N > 1 line can be audited rank by rank: this rank's own wall clock over the timed region (the
    # metric uses the MAX) and its event-timed kernel durations
    roof = rep_.get("roofline") or {}
    mine = torch.tensor([[float(rank), float(dev_index), own_wall / args.steps * 1e3, float(roof.get("fwd_ms") or 0.0),
                          float(roof.get("bwd_ms") or 0.0)]], dtype=torch.float64, device=rdev)
    allr = shard.gather_scenes(mine, world, device=rdev) if world > 1 else mine
    out["per_rank"] = [{"rank": int(r[0]), "device": int(r[1]), "ms_per_step": float(r[2]), "fwd_ms": float(r[3]), "bwd_ms": float(r[4])}
                       for r in allr.cpu().tolist()]
    out["config"]["global_batch"] = work.units_per_step * world
    out["config"]["parallelism"] = "scenes sharded x%d, no collectives" % world
    if is_gpu and devices_used != reported:
        out["devices_used"] = devices_used
    if sustained is not None:
        out["sustained"] = sustained
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out.update(work.host_side_checks())
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return out if rank == 0 else None


# ------------------------------------------------------------------------------------------------ CPU side (rank 0, N = 1)
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


def cpu_reference_quote():
    """The unmodified reference's own timing of this workload (build container; it cannot run on the GPU box)."""
    path = os.path.join(ROOT, "profiles", "r01_reference_cpu_timing.json")
    if not os.path.exists(path):
        return None
    j = json.load(open(path))
    r = j["runs"][0]
    return {"value": r["value"], "unit": r["unit"], "cores": r["cpu_threads"], "kind": "reference", "dtype": r["dtype"],
            "machine": r.get("machine", "build container"), "what": r["what"], "workload": r["workload"],
            "measured_in_this_run": False, "source": "profiles/r01_reference_cpu_timing.json (" + j["source"] + ")"}


def parity_vs_oracle(lcp_gpu, cot, x_gpu, z_gpu, s_gpu, iters_gpu, dp_gpu, n=512):
    """The metric's second half ("fwd+bwd rel-err vs ref"): the step just timed against the fp64 oracle on `n` scenes sampled
    over the batch, on IDENTICAL inputs - the fp32 LCP data the HIP assembly produced for the kernel (all assembling kernels
    share one contraction-free builder; tests/test_hip_parity.py::test_assembly_kernel_matches_oracle), solved by the oracle in
    fp64.  Fields: tests/parity.py::headline_report - err_x, contact index sets UNMASKED and on the decisive rows,
    the histogram of iteration-count differences, dl/dp on the scenes whose backward system is well posed; the same report
    tests/test_hip_headline_parity.py gates at configs[1] / [2] / [3]."""
    from oracle import pdipm_oracle as O
    from tests import parity
    B = x_gpu.shape[0]
    n = min(n, B)
    idx = torch.arange(0, B, max(1, B // n))[:n]
    di = idx.to(x_gpu.device)
    lcp64 = [None if t is None else t[di].double().cpu() for t in lcp_gpu]
    rep, _ = parity.headline_report(O, lcp64, x_gpu[di].cpu(), z_gpu[di].cpu(), s_gpu[di].cpu(), iters_gpu[di].cpu(),
                                    dp=None if dp_gpu is None else dp_gpu[di].cpu(), cot=cot[idx])
    rep["sample"] = "every %d-th scene of the batch" % max(1, B // n)
    return rep


def _quoted(name, key):
    """A committed per-configuration profile figure (rocprofv3 counters / compiler register report), newest round first."""
    for rnd in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s.json" % (rnd, name))
        if os.path.exists(path):
            j = json.load(open(path)).get(key)
            if j:
                j = dict(j)
                j.setdefault("source", "profiles/%s_%s.json" % (rnd, name))
                return j
    return None


# ------------------------------------------------------------------------------------------------ the HIP workload
class HipStackWorkload:
    """B stack scenes per rank resident in HBM; step = fused step kernel (or the dense operator) + backward."""

    def __init__(self, args, rank, dev):
        from lcp_physics_amd import scenes
        from lcp_physics_amd.lcp import lcp_backward, lcp_solve
        from lcp_physics_amd.physics import assemble_contacts, fused_step
        from lcp_physics_amd.physics.batched_world import solution_of_step
        if torch.device(dev).type != "cuda":
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        if args.bwd == "physical" and args.mode != "fused":
            raise SystemExit("--bwd physical needs --mode fused")
        self.args, self.rank, self.dev = args, rank, dev
        B = self.B = self.units_per_step = args.batch
        self.nb, self.nc = args.nbox + 1, args.nbox * args.pts
        self.nz, self.m, self.e = 3 * self.nb, 4 * self.nc, 3
        # synthetic scenes, generated on the host, then resident in HBM before any timing
        self.sc_cpu = scenes.make_stack_scenes(B=B, nbox=args.nbox, pts_per_interface=args.pts, seed=1236 + 1000 * rank,
                                               dtype=torch.float32)
        self.sc = self.sc_cpu.to(device=dev)
        g = torch.Generator().manual_seed(4321 + rank)
        self.cot_cpu = torch.randn(B, self.nz, generator=g, dtype=torch.float32)
        self.cot = self.cot_cpu.to(dev)
        self.lcp = assemble_contacts(self.sc)   # dense (Q,p,G,h,A,b,F) in HBM, built by the HIP assembly kernel
        G, A = self.lcp[2], self.lcp[4]
        self.sol = lcp_solve(*self.lcp, compute=args.compute)
        self.grads = lcp_backward(self.sol, self.cot)
        # the timed step asks for what the reference's step returns - new_v (and the moved pose): engines.py:76-77, bodies.py:80-82; the
        # multipliers stay in the workspace for the backward (fp64).  host_side_checks() repeats the call WITH z, s for the parity object
        # and requires its new_v / iteration counts to be bitwise those of the timed call.
        self.step_out = fused_step(self.sc, compute=args.compute, multipliers=False) if args.mode == "fused" else None
        torch.cuda.synchronize()
        # (the output buffers and the workspace are re-used every step, so the handle the backward takes is built once)
        self.step_sol = solution_of_step(self.sc, self.step_out, G, A, compute=args.compute) if args.mode == "fused" else None
        self.cot_v = (-self.cot).reshape(B, self.nb, 3).contiguous()          # d(loss)/d(v_new) = -d(loss)/dx
        self.pgrads = None

    def new_events(self):
        return [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def step(self, ev=None):
        from lcp_physics_amd.lcp import lcp_backward, lcp_solve
        from lcp_physics_amd.physics import fused_step
        from lcp_physics_amd.physics.batched_world import fused_step_backward
        a = self.args
        if ev is not None:
            ev[0].record()
        if a.mode == "dense":
            lcp_solve(*self.lcp, compute=a.compute, ws=self.sol.ws, out=self.sol)
            s_ = self.sol
        else:
            self.step_out = fused_step(self.sc, compute=a.compute, ws=self.step_out["ws"], out=self.step_out)
            s_ = self.step_sol
        if ev is not None:
            ev[1].record()
        if a.fwd_only:
            pass
        elif a.bwd == "physical":
            self.pgrads = fused_step_backward(self.sc, self.step_out, self.cot_v, compute=a.compute, grads=self.pgrads)
        else:
            lcp_backward(s_, self.cot, out=self.grads)
        if ev is not None:
            ev[2].record()

    def metric_name(self):
        return "sim steps/sec at batch=%dx%d contacts, %s" % (self.B, self.nc, "fwd" if self.args.fwd_only else "fwd+bwd")

    def report(self, events, world):
        from lcp_physics_amd import flops
        a, B, nb, nc, nz, m, e = self.args, self.B, self.nb, self.nc, self.nz, self.m, self.e
        events = [ev for ev in events if ev is not None]                       # (the sampled steps of the timed region)
        fwd_ms = sum(ev[0].elapsed_time(ev[1]) for ev in events) / len(events)
        bwd_ms = sum(ev[1].elapsed_time(ev[2]) for ev in events) / len(events)
        iters = (self.sol.iters if a.mode == "dense" else self.step_out["iters"]).double()
        status = (self.sol.status if a.mode == "dense" else self.step_out["status"])
        it_list = iters.cpu().tolist()
        fl_alg = float(sum(flops.flops_forward(nz, m, e, it) for it in it_list))
        # the contact-list entry points run the body-space variant of lcp_fwd_quad (nz <= 16, fp64 arithmetic; the stack scenes pin
        # their floor: ALG = 2), the dense boundary the contact-space one
        body_space = a.mode == "fused" and a.compute == "f64" and nz <= 16
        fl_exec = float(sum((flops.flops_forward_executed_body_space if body_space else flops.flops_forward_executed)(nz, nc, e, it)
                            for it in it_list))
        peak = PEAK_TFLOPS[a.compute]
        alg_bytes = (flops.bytes_fused_step(nb, nc) if a.mode == "fused" else flops.bytes_forward(nz, m, e)) * B
        key = "%s_B%d_nc%d_%s" % (a.mode, B, nc, a.compute)
        # HBM traffic and issue counters: measured separately with rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE in their own passes;
        # FETCH_SIZE doubled per MI355X_MICROARCH.md) and committed under profiles/; bench.py quotes the file that matches.
        tj = _quoted("traffic", key)
        traffic = (2 * tj["fetch_kb"] + tj["write_kb"]) * 1024.0 if tj else None
        cj = _quoted("counters", key)
        sized = body_space and (nz, e) in flops.SIZED_SHAPES
        rj = _quoted("kernel_resources", "lcp_fwd_solo_9_3_8" if (body_space and B <= 1024 and (nz, e, nc) == (9, 3, 8)) else
                     "lcp_fwd_quad_f64_fused_two_waves" if (sized and B > 4096) else
                     "lcp_fwd_quad_%s_%s" % ("f64" if a.compute == "f64" else "f32", a.mode))   # (the variant this mode runs)
        # which BASELINE.json config the flags amount to (the default run is configs[2], the one the metric is quoted on)
        cfg = {(1024, 8): "configs[1]", (4096, 16): "configs[2]", (32768, 16): "configs[3] on one GPU"}.get(
            (B, nc), "configs[3]" if (B * world, nc) == (32768, 16) else "variant")
        what = "forward only" if a.fwd_only else "forward + backward (implicit diff)"
        st = status.cpu()
        tf = lambda fl, ms: fl / (ms * 1e-3) / 1e12
        roof = {"bound": "valu_fp64" if a.compute == "f64" else "valu_fp32",
                "kernel": ("lcp_fwd_solo (one scene per wavefront: batches of at most 1024 scenes; PDIPM forward%s)" % (
                    ", fused assembly + integrate",)) if (body_space and B <= 1024) else "lcp_fwd_quad<float,%s,%s,1,%d%s> (PDIPM forward%s)" % (
                    "double" if a.compute == "f64" else "float", "true" if a.mode == "fused" else "false", 2 if body_space else 0,
                    (",%d,%d,%d,%s" % (nz, e, nc, "true" if B <= 4096 else "false")) if (body_space and (nz, e) in flops.SIZED_SHAPES) else "",
                    ", fused assembly + integrate; LCP_HINT_PINNED: the wrappers checked on the host that every scene's Je pins the floor, "
                    "the launch for other equality rows is skipped" if a.mode == "fused"
                    else "; the event-timed forward call also contains the classify launch"),
                "achieved": tf(fl_exec, fwd_ms), "peak": peak, "unit": "TFLOP/s", "frac": tf(fl_exec, fwd_ms) / peak,
                "flops": "executed",
                "executed_flops_per_launch": fl_exec,
                "executed_model": ("flops.flops_forward_executed_body_space(nz, nc, neq, iters, pinned=True; trimmed for the size-specialised shapes): per "
                                   "iteration formation of Q + G^T M^-1 G (4 nc nz (nz - neq)) + LU of nz - neq rows + 2 KKT solves + residuals" if body_space
                                   else "flops.flops_forward_executed: the reduced 2 nc contact-space system"),
                "kernel_ms": fwd_ms,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                "hbm_frac": (traffic / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "frac_algorithmic": tf(fl_alg, fwd_ms) / peak,
                "algorithmic_flops_per_launch": fl_alg,
                "note_algorithmic": "SURVEY 8d counts the reference's dense formulation (LU of nineq = %d rows per iteration); the kernel "
                                    "factors %d rows: a fraction above 1 is the algorithmic saving, not an efficiency" % (m, (nz - e) if body_space else 2 * nc),
                "fwd_ms": fwd_ms, "bwd_ms": bwd_ms, "event_timed_steps": len(events)}
        if cj:
            waves = (B + 3) // 4
            useful = fl_exec / 2.0 / waves / 64.0                               # wave-wide FMA instructions' worth of executed FLOPs
            roof.update({"valu_active": cj.get("valu_active"), "wait_frac": cj.get("wait_frac"),
                         "valu_insts_per_wave": cj.get("valu_insts_per_wave"),
                         "insts_per_useful_fma": (cj["valu_insts_per_wave"] / useful) if cj.get("valu_insts_per_wave") else None,
                         "counters_source": cj["source"]})
        if tj:
            roof["traffic_source"] = tj["source"]
        if rj:
            roof["regs"] = rj
        if not a.fwd_only:
            dense_bwd = a.bwd == "dense"
            bkey = key + ("_bwd" if dense_bwd else "_bwd_physical")
            btj = _quoted("traffic", bkey)
            btraffic = (2 * btj["fetch_kb"] + btj["write_kb"]) * 1024.0 if btj else None
            # what the backward must move: G, A, the cotangent and the fp64 iterate in, the seven dense gradients out (SURVEY 8d's
            # figure also counts Q and F as reads: lcp.py:37-64 does not touch them and neither does the kernel)
            balg = ((4 * (m * nz + e * nz + nz) + 8 * (nz + e + 2 * m) + 4 * (nz * nz + nz + m * nz + m + e * nz + e + m * m)) if dense_bwd
                    else 4 * (14 * nb + 7 * nc + 3 * nb) + 4 * (11 * nb + 6 * nc)) * B
            used = btraffic if btraffic else balg
            roof["bwd"] = {"bound": "hbm",
                           "kernel": ("lcp_bwd_quad<float,double,%s> (lcp.py:37-64: one factorisation, 1 + 2 KKT solves, the seven dense gradients)"
                                      % ("true" if body_space else "false")) if dense_bwd else "lcp_bwd_step_quad (gradients w.r.t. the physical inputs)",
                           "achieved": used / (bwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": used / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "bytes": "measured traffic" if btraffic else "algorithmic",
                           "traffic": btraffic, "algorithmic_bytes_per_launch": balg, "kernel_ms": bwd_ms,
                           "executed_flops_per_launch": flops.flops_backward_executed_body_space(nz, nc, e) * B if body_space else None}
            if btj:
                roof["bwd"]["traffic_source"] = btj["source"]
        return {
            "config": {"workload": "%s: batch=%d x %d contacts (%d-box stack, %d pts/interface; nz %d, nineq %d, "
                                   "neq %d) per GPU, fp32 I/O, LCP %s, mode=%s, bwd=%s%s"
                                   % (cfg, B, nc, a.nbox, a.pts, nz, m, e, what, a.mode, "none" if a.fwd_only else a.bwd,
                                      "; step outputs: new_v and the moved pose (engines.py:76-77, bodies.py:80-82), multipliers kept in the "
                                      "workspace in fp64" if a.mode == "fused" else ""),
                       "mean_pdipm_iters": float(iters.mean()), "nonzero_status": int((st != 0).sum()),
                       "status_bits": {name: int(((st & bit) != 0).sum()) for name, bit in
                                       (("singular_Q", 1), ("singular_S11", 2), ("singular_T", 4), ("nan", 8), ("truncated", 16))}},
            "roofline": roof,
        }

    def host_side_checks(self):
        a, B, nz = self.args, self.B, self.nz
        out = {"cpu_baseline": cpu_baseline(self.sc_cpu, self.cot_cpu, a.cpu_budget, a.fwd_only),
               "cpu_reference": cpu_reference_quote()}
        x_gpu = self.sol.x if a.mode == "dense" else -self.step_out["v_new"].reshape(B, nz)
        dp_gpu = None if (a.fwd_only or a.bwd == "physical") else self.grads[1]
        src = self.sol if a.mode == "dense" else None
        if src is None:
            # the same call with the multipliers written out (the timed one keeps them in the workspace only)
            from lcp_physics_amd.physics import fused_step
            chk = fused_step(self.sc, compute=a.compute)
            torch.cuda.synchronize()
            same = bool(torch.equal(chk["v_new"], self.step_out["v_new"]) and torch.equal(chk["iters"], self.step_out["iters"])
                        and torch.equal(chk["p_new"], self.step_out["p_new"]))
            if not same:
                raise SystemExit("bench.py: the step with multipliers differs from the timed step")
        z_gpu = src.z if src is not None else chk["z"]
        s_gpu = src.s if src is not None else chk["s"]
        it_gpu = src.iters if src is not None else self.step_out["iters"]
        out["parity"] = parity_vs_oracle(self.lcp, self.cot_cpu, x_gpu, z_gpu, s_gpu, it_gpu, dp_gpu)
        if src is None:
            out["parity"]["multipliers"] = ("z, s of a repeat of the timed call that writes them out (the timed step returns new_v and the pose "
                                            "only, as the reference's step does); new_v, pose and iteration counts of the two calls: bitwise equal")
        return out


def main(argv=None):
    args = parse(argv)
    from lcp_physics_amd import shard
    if not shard.under_launcher():
        if args.gpus > 1:
            # no launcher around us: start the N ranks ourselves, one per visible device
            try:
                rc = shard.launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv),
                                        share_devices=args.share_devices)
            except RuntimeError as ex:
                raise SystemExit("bench.py: " + str(ex))
            raise SystemExit(rc)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    run_rank(args, HipStackWorkload)


if __name__ == "__main__":
    main()

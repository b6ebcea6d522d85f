#!/usr/bin/env python
"""bench.py - the BASELINE.json metric on MI355X.

metric : sim steps/sec at batch=4096 x 16 contacts, forward + backward (implicit diff)
step   : one pass of the hot path over one batch of synthetic scenes already resident in HBM
         (one sim step = assembly + one LCP solve + integrate; fwd+bwd adds the
         implicit-differentiation backward): the fused step kernel lcp_step_fused_f32 (contact list ->
         new velocities and poses) + lcp_pdipm_backward_f32, for B = 4096 scenes per GPU (floor + 4-box
         stack, 4 contact points per interface: nz 15, nineq 64, neq 3).  `--mode dense` times the dense
         LCPFunction boundary instead (lcp_pdipm_forward_f32 on pre-assembled (Q,p,G,h,A,b,F) + backward).
         `--config N` selects BASELINE.json `configs[N]`: 1 = 1024 x 8 forward only, 2 = the default (the config the
         metric is quoted on), 3 = 32768 x 16 over the ranks, 4 = 4096 x 64 contacts (the ten-box pile, nineq 256:
         lcp_step_fused_f32 -> lcp_primal_kernel<30, ..., PIN>, backward lcp_step_backward_f32 - gradients w.r.t. the
         physical inputs; the dense gradients of a 256-row LCP are 302 KB per scene and no world asks for them).
launch : two eager launches per step (one ctypes call each; `--launch eager`, the default).  `--launch graph` captures the
         forward + backward pair ONCE into a HIP graph (the library is capture-safe: it launches on the caller's stream and keeps
         no host state) and replays it K times: measured SLOWER on this stack (0.1006 against 0.0947 ms per step at the headline,
         profiles/r04_ab_launch.txt - the eager queue already runs the two kernels back to back, 94.7 us per step against 93.6 us
         of kernel time), so it is the companion figure `graph_ms_per_step`, not the timed form.
N GPUs : one process per GPU, every rank owns its own 4096 scenes (weak scaling, config 3 = 8 x 4096); no
         collective on the solve path - torch.distributed (RCCL) is only used for the barriers and the
         MAX-over-ranks wall time.  `python bench.py --gpus N` started WITHOUT a launcher starts the N ranks
         itself (torch.distributed.run on 127.0.0.1, one per visible device) and refuses when fewer than N
         devices are visible; under torchrun (the driver's form) --gpus must equal WORLD_SIZE.  `n_gpus` in
         the output is the number of ranks that reported, on distinct devices.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      - dominant kernel (the forward PDIPM kernel).  `achieved` / `frac` = the FLOPs the kernel really EXECUTES
                  (lcp_physics_amd/flops.py, with the iteration counts the kernel reports) divided by its average launch
                  duration - HIP events on the launch stream around >= 128 eager launches right after the timed region
                  (`event_timed_steps`; events inside the timed region would cost the stream 2-3 us each) - over the FP64
                  VECTOR rate (the kernels issue no MFMA: `bound` says "valu_fp64").  `frac_necessary` = the same with the
                  formation counted SPARSE (a contact touches two bodies: 6 x 6 block): what a kernel that multiplied no
                  structural zero would execute.  `frac_algorithmic` = SURVEY.md §8d's count of the REFERENCE's dense
                  formulation over the same time and peak: it exceeds 1 because the kernel factors nz - neq rows where the
                  reference factors nineq - a different algorithm with the same answers, not an efficiency.
                  `insts_per_useful_fma` cross-checks the executed model against the SQ_INSTS_VALU counter; `traffic` = HBM
                  bytes per launch from the FETCH_SIZE / WRITE_SIZE counters; counters and registers are quoted from the
                  committed rocprofv3 / compiler reports under profiles/ (bench.py cannot run --pmc itself).
                  `roofline.bwd` = the backward kernel: the dense backward against the HBM roofline (it is bound by the
                  21.6 KB of dense gradients it writes per scene), the physical one against the FP64 vector rate.
  companions    - (rank 0, N = 1) the same workload timed three more ways, K steps each (best of two repeats): `graph` (one HIP-graph replay per step),
                  `general_kernel` (lcp_solve_dynamics_f32 with a contact count per scene: the instantiation a ContactWorld
                  gets) and `with_multipliers` (z, s, y written out every step: the form rounds 1 and 2 timed).
  cpu_baseline  - the oracle (a port, torch CPU fp64) timed on this host's cores on the same workload (rank 0,
                  N=1 only);  cpu_reference - the UNMODIFIED reference's own timing of the workload (newest
                  profiles/r*_reference_cpu_timing.json; /root/reference does not exist on the GPU box).
  parity        - the metric's second half: tests/parity.py::headline_report on 512 scenes sampled over the batch.
  sustained     - the same step repeated for >= 1 s after the timed region (independent evidence of GPU work).
"""
import argparse
import glob
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}   # MI355X FP64 / FP32 vector = matrix rates (datasheet; MI355X_MICROARCH.md lists FP32)
HBM_PEAK_GBS = 8000.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: enough warm-up for the clocks to settle (the first tens of launches of a cold device run ~10 % slower) and a
    # timed region of ~40 ms; the whole default run still takes seconds
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[N]: sets --batch / --nbox / --pile / --fwd-only (2 = the default run)")
    ap.add_argument("--batch", type=int, default=None, help="scenes per GPU (default 4096)")
    ap.add_argument("--nbox", type=int, default=None)
    ap.add_argument("--pts", type=int, default=4)
    ap.add_argument("--pile", action="store_true", help="BASELINE configs[4]: the ten-box pyramid, 64 contacts (nz 33, nineq 256)")
    ap.add_argument("--compute", default="f64", choices=["f64", "f32"])
    ap.add_argument("--mode", default="fused", choices=["dense", "fused"])
    ap.add_argument("--contact-space", action="store_true",
                    help="--mode dense: LCP_PATH_CONTACT_SPACE, the contact-space factorisation (the default until round 3) instead of the body-space kernels")
    ap.add_argument("--bwd", default=None, choices=["dense", "physical"],
                    help="backward timed in the step: 'dense' = LCPFunction.backward (7 dense gradients, lcp.py:37-64); "
                         "'physical' (fused mode only) = lcp_step_backward_f32, gradients w.r.t. the physical inputs "
                         "(default: dense for the stacks, physical for the pile)")
    ap.add_argument("--fwd-only", action="store_true",
                    help="time the forward only (BASELINE configs[1] is forward-only); the default is the headline fwd+bwd")
    ap.add_argument("--launch", default="eager", choices=["graph", "eager"],
                    help="how the timed region issues a step: two eager launches (default: measured faster, see `graph_ms_per_step`), or "
                         "one replay of a captured HIP graph (fwd + bwd)")
    ap.add_argument("--spinup", type=float, default=0.3,
                    help="seconds of untimed launches while the workload is built, before the W warm-up steps: the first ~0.2 s of "
                         "launches on a fresh process run at ramping clocks, which a 2 ms timed region (--steps 20) would otherwise measure")
    ap.add_argument("--event-samples", type=int, default=128, help="eager launches bracketed by HIP events for roofline.kernel_ms")
    ap.add_argument("--event-group", type=int, default=8, help="launches of one kind between two event records (1: every launch bracketed by itself)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-companions", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU work for the baseline")
    ap.add_argument("--sustain", type=float, default=1.0, help="seconds of the untimed sustained loop after the timed steps (0 = off)")
    ap.add_argument("--share-devices", action="store_true",
                    help="TESTING AID: let several ranks use one GPU (gloo instead of RCCL); the line then says devices_used < n_gpus")
    a = ap.parse_args(argv)
    if a.config == 1:
        a.batch, a.nbox, a.fwd_only = a.batch or 1024, a.nbox or 2, True
    elif a.config == 3:
        a.batch, a.nbox = a.batch or max(1, 32768 // max(1, a.gpus)), a.nbox or 4
    elif a.config == 4:
        a.pile = True
    a.batch = a.batch or 4096
    if a.pile:
        a.nbox, a.pts = 10, 4
        if a.mode == "fused":
            if a.bwd == "dense":
                raise SystemExit("bench.py: the backward of the pile's contact-list step is lcp_step_backward_f32 (--bwd physical); "
                                 "--mode dense times the seven dense gradients")
            a.bwd = "physical"
        else:
            # the dense LCPFunction boundary at nineq 256: lcp_pdipm_forward_f32 on 302 KB of (Q, p, G, h, A, b, F) per scene and the
            # seven dense gradients of lcp_pdipm_backward_f32 (another 302 KB per scene)
            if a.bwd == "physical":
                raise SystemExit("bench.py: --bwd physical needs --mode fused")
            a.bwd = "dense"
    a.nbox = a.nbox or 4
    a.bwd = a.bwd or "dense"
    return a


# ------------------------------------------------------------------------------------------------ rank protocol
SPINUP_MIN_STEPS = 1500          # (see HipWorkload.__init__: the runtime's one-off stall lies behind this many steps at every workload measured)
SPINUP_MAX_S = 3.0               # ... but never longer than this (steps of 10-100 ms: the variants with 20 bodies, the contact-space dense boundary of configs[4])


def timed_steps(work, steps, warmup, sync):
    """The contract's timing rule: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier +
    device synchronisation on both sides.  Returns (this rank's wall time incl. the closing barrier, its own wall time before it
    waits for the others).  Nothing but `work.step()` runs inside the timed region (no event records)."""
    from lcp_physics_amd import shard
    for _ in range(warmup):
        work.step()
    sync()
    shard.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        work.step()
    sync()
    own = time.perf_counter() - t0                   # this rank's K steps, device work included, before it waits for the others
    shard.barrier()
    wall = time.perf_counter() - t0
    return wall, own


def run_rank(args, make_work, device=None):
    """Everything one rank does.  `make_work(args, rank, device)` builds the workload (HipWorkload below; the CPU
    test of the N > 1 protocol passes a stand-in whose step launches nothing).  Returns the JSON object on rank 0."""
    from lcp_physics_amd import shard
    rank, local_rank, world = shard.env_rank()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    # (device given: the CPU test of the protocol - gloo only.  Otherwise RCCL is tried and gloo carries the run if it does not come up,
    #  e.g. with --share-devices, where two ranks on one GPU are something RCCL refuses)
    shard.init_process_group(backend="gloo" if device is not None else None)
    dev = device if device is not None else shard.rank_device(local_rank, share_devices=args.share_devices)
    is_gpu = torch.device(dev).type == "cuda"
    if is_gpu:
        torch.cuda.set_device(dev)
    rdev = shard.reduce_device(dev)
    sync = torch.cuda.synchronize if is_gpu else (lambda: None)
    work = make_work(args, rank, dev)
    own_wall_all, own_wall = timed_steps(work, args.steps, args.warmup, sync)
    wall = shard.max_over_ranks(own_wall_all, device=rdev)
    # kernel durations: HIP events around eager launches, outside the timed region
    samples = work.sample_kernels(args.event_samples) if hasattr(work, "sample_kernels") else None
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
    # what a step costs THIS rank's host thread with the device idle in front of it: the launches of a few steps issued back to back, the
    # clock stopped before the device is waited for.  With N ranks on one host these are N threads doing this concurrently; the step
    # cannot go faster than the slowest of them issues it - `host_bound_ceiling` below is that bound in the metric's unit, so that a
    # scaling result below 7 x at N = 8 can be read off the line: kernels slower (per_rank fwd_ms / bwd_ms) or hosts slower (host_us_per_step)
    sync()
    n_issue = 20 if is_gpu else 3
    t0 = time.perf_counter()
    for _ in range(n_issue):
        work.step()
    host_us = (time.perf_counter() - t0) / n_issue * 1e6
    sync()
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
    rep_ = work.report(samples, world)                # (every rank: its own kernel timings go into `per_rank`)
    out.update(rep_)
    # per-rank record, so that an N > 1 line can be audited rank by rank: this rank's own wall clock over the timed region (the
    # metric uses the MAX) and its event-timed kernel durations
    roof = rep_.get("roofline") or {}
    if roof.get("fwd_ms") is not None:
        # the event-bracketed launches are a little LONGER than a step of the timed region (an event record on the stream between two
        # kernels costs it a microsecond or two; rocprofv3's kernel-trace durations lie in between): fractions priced on `kernel_ms` /
        # `fwd_ms` / `bwd_ms` are therefore slightly conservative, never optimistic
        roof["kernel_ms_sum"] = float(roof["fwd_ms"]) + float(roof.get("bwd_ms") or 0.0)
        roof["kernel_ms_sum_over_step_ms"] = roof["kernel_ms_sum"] / (wall / args.steps * 1e3)
        roof["kernel_ms_note"] = ("fwd_ms + bwd_ms (HIP events around groups of launches, after the timed region) vs ms_per_step (wall clock over K "
                                  "un-instrumented steps of forward + backward): what is left above 1 is the event records' share and the "
                                  "difference between back-to-back launches of one kernel and the alternating pair; achieved / frac are conservative")
    mine = torch.tensor([[float(rank), float(dev_index), own_wall / args.steps * 1e3, float(roof.get("fwd_ms") or 0.0),
                          float(roof.get("bwd_ms") or 0.0), host_us]], dtype=torch.float64, device=rdev)
    allr = shard.gather_scenes(mine, world, device=rdev) if world > 1 else mine
    out["per_rank"] = [{"rank": int(r[0]), "device": int(r[1]), "ms_per_step": float(r[2]), "fwd_ms": float(r[3]), "bwd_ms": float(r[4]),
                        "host_us_per_step": float(r[5])} for r in allr.cpu().tolist()]
    out["host_us_per_step"] = max(r["host_us_per_step"] for r in out["per_rank"])
    out["host_bound_ceiling"] = {"value": work.units_per_step * world / (out["host_us_per_step"] * 1e-6), "unit": "sim steps/s",
                                 "note": "all ranks' units per step / the slowest rank's host time to ISSUE a step (its launches back to back, device not "
                                         "waited for): the value cannot exceed this however fast the kernels are"}
    out["config"]["global_batch"] = work.units_per_step * world
    out["config"]["parallelism"] = "scenes sharded x%d, no collectives" % world
    out["config"]["control_plane"] = shard.control_plane()          # what carried the barriers / the MAX of the wall times
    if is_gpu and devices_used != reported:
        out["devices_used"] = devices_used
    if sustained is not None:
        out["sustained"] = sustained
    if rank == 0 and world == 1 and not args.no_companions and hasattr(work, "companions"):
        out.update(work.companions(sync))
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
    big = sc_cpu.nc > 32                                # (the piles: 256 x 256 dense systems per scene)
    if big:
        # torch.set_num_threads() (whatever the value) turns on MKL's own threading inside ATen's batch-parallel LU, and the batched
        # 256 x 256 getrf then fails in DLASWP or hangs (torch 2.10 + oneMKL 2024.2; the 64 x 64 systems of the stacks take MKL's
        # sequential path and are not affected): the piles run with the thread settings the process started with
        threads = torch.get_num_threads()
    else:
        threads = max(1, min(os.cpu_count() or 1, 16))  # tiny batched ops: more threads only add overhead
        torch.set_num_threads(threads)

    def run(n):
        sub = sc_cpu.slice(0, n).to(dtype=torch.float64)
        lcp = O.assemble_lcp(*sub.assembly_args())
        t0 = time.perf_counter()
        sol = O.lcp_forward(*lcp)
        if not fwd_only:
            O.lcp_backward(sol, *lcp, cot[:n].double())
        return time.perf_counter() - t0

    run(8 if big else 32)                               # warm-up
    n0 = 32 if big else 128
    cal = run(n0)
    sample = int(max(n0, min(sc_cpu.B, n0 * budget_s / max(cal, 1e-3))))
    total, passes = 0.0, 0
    while total < budget_s * 0.66 and passes < 64:      # repeat passes over the sample up to ~10 s of CPU work
        total += run(sample)
        passes += 1
    return {"value": sample * passes / total, "unit": "sim steps/s", "cores": threads, "kind": "port",
            "sample": "%d of the same scenes x %d passes, %s, vectorised torch fp64 oracle, %.2f s"
                      % (sample, passes, "forward only" if fwd_only else "fwd+bwd", total)}


def cpu_reference_quote(nc=16):
    """The unmodified reference's own timing of this workload: the BEST run (over thread counts and dtypes of the reference's own
    choice, fp64 first - its default, utils.py:34) of the newest committed profiles/r*_reference_cpu_timing.json whose workload matches.
    /root/reference does not exist on the GPU box, so the driver's run can only quote; see `cpu_reference_live` for the run that measures."""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_cpu_timing.json")), reverse=True)
    for path in paths:
        j = json.load(open(path))
        runs = [r for r in j["runs"] if int(r.get("nc", 16)) == nc]
        if not runs:
            continue
        f64 = [r for r in runs if r["dtype"] == "float64"] or runs
        r = max(f64, key=lambda q: q["value"])
        return {"value": r["value"], "unit": r["unit"], "cores": r["cpu_threads"], "kind": "reference", "dtype": r["dtype"],
                "machine": r.get("machine", j.get("machine", "build container")), "what": r.get("what"), "workload": r.get("workload"),
                "thread_counts_tried": sorted({q["cpu_threads"] for q in f64}),
                "measured_in_this_run": False, "source": os.path.relpath(path, ROOT) + " (" + j["source"] + ")"}
    return None


def cpu_reference_live(nc, quote):
    """MEASUREMENT RUNS ONLY: when a copy of the reference's python package is staged for this call (LCP_REFERENCE_ROOT, see
    tools/stage_reference.sh - never on the driver's box), the unmodified reference is timed HERE, in this process, on a 256-scene sample
    of the same workload at the thread count the committed sweep found best.  Returns None when nothing is staged."""
    root = os.environ.get("LCP_REFERENCE_ROOT")
    if not root or not os.path.isdir(os.path.join(root, "lcp_physics")):
        return None
    from oracle import time_reference as TR
    before = torch.get_num_threads()
    try:
        threads = int(quote["cores"]) if quote else 8
        r = TR.time_reference(batch=64 if nc > 32 else 256, dtype="float64", reps=1, pile=nc > 32, threads=threads,
                              machine="this host (%d cores), inside the bench run" % (os.cpu_count() or 0))
    finally:
        torch.set_num_threads(before)
    return {"value": r["value"], "unit": r["unit"], "cores": r["cpu_threads"], "kind": "reference", "dtype": r["dtype"], "machine": r["machine"],
            "what": r["what"], "workload": r["workload"], "sample": "%d scenes, one pass, %.1f s" % (r["batch"], r["fwd_s"] + r["bwd_s"]),
            "measured_in_this_run": True, "source": "LCP_REFERENCE_ROOT (a staged copy of the reference's python package; measurement call only)"}


_SRC_SHA = None


def _quoted(name, key):
    """A committed per-configuration profile figure (rocprofv3 counters / compiler register report), newest round first.  Every such file
    is stamped with the sha256 of the kernel sources it was measured on (lcp_physics_amd/srchash.py); a figure whose stamp is not the one
    of the sources this run executes comes back with "counters_stale": true - quoted for orientation, not as a measurement of this build."""
    global _SRC_SHA
    if _SRC_SHA is None:
        from lcp_physics_amd.srchash import source_sha256
        _SRC_SHA = source_sha256()
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s.json" % (rnd, name))
        if os.path.exists(path):
            doc = json.load(open(path))
            j = doc.get(key)
            if j:
                j = dict(j)
                j.setdefault("source", "profiles/%s_%s.json" % (rnd, name))
                j["counters_stale"] = doc.get("source_sha256") != _SRC_SHA
                if j["counters_stale"]:
                    j["stale_note"] = ("measured on other kernel sources than this run's (source_sha256 %s in the file, %s here): re-run tools/collect_round.sh"
                                       % (str(doc.get("source_sha256"))[:12], _SRC_SHA[:12]))
                return j
    return None


# ------------------------------------------------------------------------------------------------ the HIP workload
class HipWorkload:
    """B scenes per rank resident in HBM (4-box stacks, or the ten-box piles of configs[4]); step = fused step kernel (or the
    dense operator) + backward, replayed from one HIP graph or issued eagerly."""

    def __init__(self, args, rank, dev):
        from lcp_physics_amd import scenes
        from lcp_physics_amd.lcp import lcp_backward, lcp_solve
        from lcp_physics_amd.physics import assemble_contacts, fused_step
        from lcp_physics_amd.physics.batched_world import fused_step_backward, solution_of_step
        if torch.device(dev).type != "cuda":
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        if args.bwd == "physical" and args.mode != "fused":
            raise SystemExit("--bwd physical needs --mode fused")
        self.args, self.rank, self.dev = args, rank, dev
        self.pile = bool(args.pile)
        B = self.B = self.units_per_step = args.batch
        # synthetic scenes, generated on the host, then resident in HBM before any timing
        if self.pile:
            self.sc_cpu = scenes.make_pile_scenes(B=B, seed=5 + 1000 * rank, dtype=torch.float32)
        else:
            self.sc_cpu = scenes.make_stack_scenes(B=B, nbox=args.nbox, pts_per_interface=args.pts, seed=1236 + 1000 * rank,
                                                   dtype=torch.float32)
        self.nb, self.nc = self.sc_cpu.nb, self.sc_cpu.nc
        self.nz, self.m, self.e = 3 * self.nb, 4 * self.nc, 3
        self.sc = self.sc_cpu.to(device=dev)
        g = torch.Generator().manual_seed(4321 + rank)
        self.cot_cpu = torch.randn(B, self.nz, generator=g, dtype=torch.float32)
        self.cot = self.cot_cpu.to(dev)
        self.cot_v = (-self.cot).reshape(B, self.nb, 3).contiguous()          # d(loss)/d(v_new) = -d(loss)/dx
        self.lcp = self.sol = self.grads = self.step_sol = self.pgrads = None
        if not self.pile or args.mode == "dense":
            # dense (Q,p,G,h,A,b,F) in HBM, built by the HIP assembly kernel (the dense backward reads G and A; --mode dense solves it)
            self.lcp = assemble_contacts(self.sc)
            self.sol = lcp_solve(*self.lcp, compute=args.compute, path="big" if args.contact_space else "auto")
            self.grads = lcp_backward(self.sol, self.cot)
        # the timed step asks for what the reference's step returns - new_v (and the moved pose): engines.py:76-77, bodies.py:80-82; the
        # multipliers stay in the workspace for the backward (fp64).  host_side_checks() repeats the call WITH z, s for the parity object
        # and requires its new_v / iteration counts to be bitwise those of the timed call.
        self.step_out = fused_step(self.sc, compute=args.compute, multipliers=False) if args.mode == "fused" else None
        torch.cuda.synchronize()
        # (the output buffers and the workspace are re-used every step, so the handle the backward takes is built once)
        if args.mode == "fused" and not self.pile:
            self.step_sol = solution_of_step(self.sc, self.step_out, self.lcp[2], self.lcp[4], compute=args.compute)
        if args.bwd == "physical" and not args.fwd_only:
            self.pgrads = fused_step_backward(self.sc, self.step_out, self.cot_v, compute=args.compute)
        torch.cuda.synchronize()
        self.graph = None
        self.spinup_s = 0.0
        self.spinup_steps = 0
        if args.spinup > 0:                                # (device clocks settle: part of building the workload, not of the W warm-up steps)
            # ... and so does the HIP runtime: every process stalls ONCE for 30-40 ms somewhere between its 700th and its 1700th kernel launch
            # (profiles/r06_step_hiccups.txt: one slow group per process, at every workload) - with the 0.5 ms steps of the 32768-scene lines
            # 0.3 s of spin-up ended before it and the stall landed in some timed regions (r06_ab_timed_region.txt).  The spin-up therefore also
            # lasts at least SPINUP_MIN_STEPS steps (capped at SPINUP_MAX_S seconds for workloads whose steps take tens of milliseconds).
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < args.spinup or (self.spinup_steps < SPINUP_MIN_STEPS and time.perf_counter() - t0 < SPINUP_MAX_S):
                for _ in range(50):
                    self.eager_step()
                self.spinup_steps += 50
                torch.cuda.synchronize()
            self.spinup_s = time.perf_counter() - t0
        if args.launch == "graph":
            self.graph = self.capture()

    def capture(self):
        for _ in range(3):
            self.eager_step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.eager_step()
        torch.cuda.synchronize()
        return g

    # ---- one step
    def forward(self):
        from lcp_physics_amd.lcp import lcp_solve
        from lcp_physics_amd.physics import fused_step
        a = self.args
        if a.mode == "dense":
            lcp_solve(*self.lcp, compute=a.compute, ws=self.sol.ws, out=self.sol, path="big" if a.contact_space else "auto")
        else:
            self.step_out = fused_step(self.sc, compute=a.compute, ws=self.step_out["ws"], out=self.step_out, multipliers=False)

    def backward(self):
        from lcp_physics_amd.lcp import lcp_backward
        from lcp_physics_amd.physics.batched_world import fused_step_backward
        a = self.args
        if a.fwd_only:
            return
        if a.bwd == "physical":
            self.pgrads = fused_step_backward(self.sc, self.step_out, self.cot_v, compute=a.compute, grads=self.pgrads)
        else:
            lcp_backward(self.sol if a.mode == "dense" else self.step_sol, self.cot, out=self.grads)

    def eager_step(self):
        self.forward()
        self.backward()

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.eager_step()

    def sample_kernels(self, n):
        """HIP events on the launch stream around eager launches, `n` of each kind: G forwards back to back between two events, then G
        backwards between two events (G = --event-group, default 8) - an event record costs the stream 1-2 us, which bracketing every
        single launch (rounds 1-4) booked as kernel time: 6-8 % of a 50 us kernel (`kernel_ms_sum_over_step_ms` was 1.06-1.08).  The first
        group is dropped: the queue is empty when it arrives."""
        g = max(1, int(self.args.event_group))
        groups = max(2, (max(2, n) + g - 1) // g + 1)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(groups)]
        torch.cuda.synchronize()
        for ev in evs:
            ev[0].record()
            for _ in range(g):
                self.forward()
            ev[1].record()
            for _ in range(g):
                self.backward()
            ev[2].record()
        torch.cuda.synchronize()
        return [(ev[0].elapsed_time(ev[1]) / g, ev[1].elapsed_time(ev[2]) / g) for ev in evs[1:]]

    def _time(self, fn, sync):
        a = self.args
        best = None
        for _ in range(2):                                 # (best of two repeats of K steps: a companion figure, not the metric)
            for _ in range(min(max(a.warmup, 2), 10)):
                fn()
            sync()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                fn()
            sync()
            dt = (time.perf_counter() - t0) / a.steps
            best = dt if best is None else min(best, dt)
        return best

    def companions(self, sync):
        """The same workload timed other ways (K steps each, this rank's clock): see the module docstring."""
        from lcp_physics_amd.lcp import lcp_backward
        from lcp_physics_amd.physics import fused_step
        from lcp_physics_amd.physics.batched_world import fused_step_backward, solution_of_step, solve_dynamics
        from lcp_physics_amd.physics.contacts import ContactBuffers
        a, B = self.args, self.B
        out = {}
        rate = lambda dt: {"ms_per_step": dt * 1e3, "value": B / dt, "unit": "sim steps/s"}
        if self.graph is not None:
            dt = self._time(self.eager_step, sync)
            out["eager_ms_per_step"] = dt * 1e3
            out["eager_value"] = B / dt
        else:
            g = self.capture()
            dt = self._time(g.replay, sync)
            out["graph_ms_per_step"] = dt * 1e3
            out["graph_value"] = B / dt
        if a.mode != "fused":
            return out
        sc = self.sc
        # (1) per-scene contact counts: lcp_solve_dynamics_f32 - what ContactWorld.step() calls (LCP_HINT_PINNED checked on the host)
        cb = ContactBuffers(B, self.nb, self.nc, self.dev)
        cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2
        count = torch.full((B,), self.nc, dtype=torch.int32, device=self.dev)
        st = {"o": None}

        def sd():
            o = st["o"]
            st["o"] = solve_dynamics(B, self.nb, self.nc, self.e, count, sc.Mdiag, sc.v, sc.f, sc.rest, sc.fric, cb, sc.Je, sc.dt,
                                     compute=a.compute, ws=None if o is None else o["ws"], out=o, pinned=True)
        sd()
        sync()
        same = bool(torch.equal(st["o"]["v_new"], self.step_out["v_new"]))
        if a.fwd_only:
            gen = sd
        elif a.bwd == "physical":
            pg = {"g": None}

            def gen():
                sd()
                pg["g"] = fused_step_backward(sc, st["o"], self.cot_v, compute=a.compute, grads=pg["g"])
        else:
            gsol = solution_of_step(sc, st["o"], self.lcp[2], self.lcp[4], compute=a.compute)
            gg = lcp_backward(gsol, self.cot)

            def gen():
                sd()
                lcp_backward(gsol, self.cot, out=gg)
        r = rate(self._time(gen, sync))
        r.update({"launch": "eager", "new_v_bitwise_equal_to_the_timed_kernel": same,
                  "what": "lcp_solve_dynamics_f32 with c_count[B] (= the full list here) + the same backward: the run-time-count "
                          "instantiation a ContactWorld gets"})
        out["general_kernel"] = r
        out["general_kernel_value"] = r["value"]
        # (2) multipliers written out every step (z, s, y to HBM in fp32): the step rounds 1-2 timed
        mo = {"o": fused_step(sc, compute=a.compute)}
        msol = None if (self.pile or a.bwd == "physical" or a.fwd_only) else \
            solution_of_step(sc, mo["o"], self.lcp[2], self.lcp[4], compute=a.compute)
        mg = {"g": None if msol is None else lcp_backward(msol, self.cot), "p": None}

        def withm():
            mo["o"] = fused_step(sc, compute=a.compute, ws=mo["o"]["ws"], out=mo["o"])
            if a.fwd_only:
                return
            if a.bwd == "physical":
                mg["p"] = fused_step_backward(sc, mo["o"], self.cot_v, compute=a.compute, grads=mg["p"])
            else:
                lcp_backward(msol, self.cot, out=mg["g"])
        r = rate(self._time(withm, sync))
        r.update({"launch": "eager", "what": "the same step with z, s, y written out (multipliers=True)"})
        out["with_multipliers"] = r
        return out

    def metric_name(self):
        return "sim steps/sec at batch=%dx%d contacts, %s" % (self.B, self.nc, "fwd" if self.args.fwd_only else "fwd+bwd")

    def report(self, samples, world):
        from lcp_physics_amd import flops
        a, B, nb, nc, nz, m, e = self.args, self.B, self.nb, self.nc, self.nz, self.m, self.e
        fwd_ms = sum(s[0] for s in samples) / len(samples)
        bwd_ms = sum(s[1] for s in samples) / len(samples)
        iters = (self.sol.iters if a.mode == "dense" else self.step_out["iters"]).double()
        status = (self.sol.status if a.mode == "dense" else self.step_out["status"])
        it_list = iters.cpu().tolist()
        fl_alg = float(sum(flops.flops_forward(nz, m, e, it) for it in it_list))
        # the contact-list entry points run the body-space variant of lcp_fwd_quad (nz <= 16, fp64 arithmetic; the stack scenes pin
        # their floor: ALG = 2), the one-wave-per-scene lcp_primal_kernel for the piles, the dense boundary the contact-space one
        body_space = a.compute == "f64" and nz <= 16 and not (a.mode == "dense" and a.contact_space)
        primal = self.pile and a.compute == "f64" and not (a.mode == "dense" and a.contact_space)
        dense_pile = self.pile and a.mode == "dense"      # the dense boundary at nineq 256: bound by the 302 KB per scene it reads / writes
        model = (flops.flops_forward_executed_primal if primal else
                 flops.flops_forward_executed_body_space if body_space else flops.flops_forward_executed)
        fl_exec = float(sum(model(nz, nc, e, it) for it in it_list))
        fl_nec = float(sum(flops.flops_forward_executed_primal(nz, nc, e, it, True) for it in it_list)) if (body_space or primal) else None
        peak = PEAK_TFLOPS[a.compute]
        alg_bytes = (flops.bytes_fused_step(nb, nc) if a.mode == "fused" else flops.bytes_forward(nz, m, e)) * B
        key = "%s%s_B%d_nc%d_%s" % (a.mode, "_cs" if (a.mode == "dense" and a.contact_space) else "", B, nc, a.compute)
        # HBM traffic and issue counters: measured separately with rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE in their own passes;
        # FETCH_SIZE doubled per MI355X_MICROARCH.md) and committed under profiles/; bench.py quotes the file that matches.
        tj = _quoted("traffic", key)
        traffic = (2 * tj["fetch_kb"] + tj["write_kb"]) * 1024.0 if tj else None
        cj = _quoted("counters", key)
        sized = body_space and (nz, e) in flops.SIZED_SHAPES
        rj = _quoted("kernel_resources", "lcp_big_kernel_64_dense_fwd" if (dense_pile and not primal) else
                     "lcp_primal_kernel_dense_fwd" if dense_pile else
                     "lcp_primal_kernel_30_pinned_fwd" if primal else
                     "lcp_fwd_quad_f64_dense_contact_space" if (a.mode == "dense" and a.contact_space) else
                     "lcp_fwd_solo_9_3_8" if (body_space and B <= 1024 and (nz, e, nc) == (9, 3, 8)) else
                     "lcp_fwd_quad_f64_fused_two_waves" if (sized and B > 4096) else
                     "lcp_fwd_quad_%s_%s" % ("f64" if a.compute == "f64" else "f32", a.mode))   # (the variant this mode runs)
        # which BASELINE.json config the flags amount to (the default run is configs[2], the one the metric is quoted on)
        cfg = {(1024, 8): "configs[1]", (4096, 16): "configs[2]", (32768, 16): "configs[3] on one GPU", (4096, 64): "configs[4]"}.get(
            (B, nc), "configs[3]" if (B * world, nc) == (32768, 16) else "variant")
        what = "forward only" if a.fwd_only else "forward + backward (implicit diff)"
        st = status.cpu()
        tf = lambda fl, ms: fl / (ms * 1e-3) / 1e12
        if dense_pile and not primal:
            kname = ("lcp_classify_big + lcp_big_kernel<64, false, DENSE> (LCP_PATH_CONTACT_SPACE: the reference's T of pdipm.py:414-454 reduced to "
                     "2 nc = 128 rows, one 256-thread workgroup per scene, blocked LU on v_mfma_f64_16x16x4_f64, factors in LDS)")
            emodel = "flops.flops_forward_executed: the reduced 2 nc contact-space system, dense arithmetic"
        elif dense_pile:
            kname = ("lcp_classify_big + lcp_primal_kernel<..., DENSE> (one wavefront per scene, body space: rows of G read from the dense "
                     "tensors; the event-timed call also holds the lcp_big_kernel / generic launches that find no scene)")
            emodel = ("flops.flops_forward_executed_primal(nz, nc, neq, iters, pinned=True): per iteration the SPARSE formation (6 x 6 block per "
                      "contact) + LU of nz - neq rows + 2 KKT solves + residuals + step lengths")
        elif primal:
            kname = ("lcp_primal_kernel<30, false, false, 4, PIN = 3> (one wavefront per scene, body space: the 30 free coordinates' system; "
                     "fused assembly + integrate; LCP_HINT_PINNED)")
            emodel = ("flops.flops_forward_executed_primal(nz, nc, neq, iters, pinned=True): per iteration the SPARSE formation (6 x 6 block per "
                      "contact) + LU of nz - neq rows + 2 KKT solves + residuals + step lengths")
        elif body_space and B <= 1024:
            kname = "lcp_fwd_solo (one scene per wavefront: batches of at most 1024 scenes; PDIPM forward, fused assembly + integrate)"
            emodel = None
        else:
            kname = "lcp_fwd_quad<float,%s,%s,1,%d%s> (PDIPM forward%s)" % (
                "double" if a.compute == "f64" else "float", "true" if a.mode == "fused" else "false", 2 if body_space else 0,
                (",%d,%d,%d,%s" % (nz, e, nc, "true" if B <= 4096 else "false")) if (body_space and (nz, e) in flops.SIZED_SHAPES) else "",
                ", fused assembly + integrate; LCP_HINT_PINNED: the wrappers checked on the host that every scene's Je pins the floor, "
                "the launch for other equality rows is skipped" if a.mode == "fused"
                else "; the event-timed forward call also contains the classify launch (and, in body space, the general kernel's launch "
                     "behind the pinned one: it finds no scene)")
            emodel = None
        if emodel is None:
            emodel = ("flops.flops_forward_executed_body_space(nz, nc, neq, iters, pinned=True; trimmed for the size-specialised shapes): per "
                      "iteration formation of Q + G^T M^-1 G (4 nc nz (nz - neq)) + LU of nz - neq rows + 2 KKT solves + residuals" if body_space
                      else "flops.flops_forward_executed: the reduced 2 nc contact-space system")
        roof = {"bound": "valu_fp64" if a.compute == "f64" else "valu_fp32",
                "kernel": kname,
                "achieved": tf(fl_exec, fwd_ms), "peak": peak, "unit": "TFLOP/s", "frac": tf(fl_exec, fwd_ms) / peak,
                "flops": "executed",
                "executed_flops_per_launch": fl_exec,
                "executed_model": emodel,
                "frac_necessary": (tf(fl_nec, fwd_ms) / peak) if fl_nec else None,
                "necessary_flops_per_launch": fl_nec,
                "necessary_model": "flops.flops_forward_executed_primal: the same solve with the formation and the J v / J^T w products "
                                   "counted over the contact's two bodies only (no structural zero multiplied)" if fl_nec else None,
                "kernel_ms": fwd_ms,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                "hbm_frac": (traffic / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "frac_algorithmic": tf(fl_alg, fwd_ms) / peak,
                "algorithmic_flops_per_launch": fl_alg,
                "note_algorithmic": "SURVEY 8d counts the reference's dense formulation (LU of nineq = %d rows per iteration); the kernel "
                                    "factors %d rows: a fraction above 1 is the algorithmic saving, not an efficiency"
                                    % (m, (nz - e) if (body_space or primal) else 2 * nc),
                "fwd_ms": fwd_ms, "bwd_ms": bwd_ms, "event_timed_steps": len(samples) * max(1, int(a.event_group)),
                "event_timing": "HIP events on the launch stream around groups of %d eager launches of one kind, right after the timed region"
                                % max(1, int(a.event_group))}
        if dense_pile:
            # 302 KB of dense tensors in per scene against ~0.6 MFLOP executed: the forward CALL (classification included) is priced
            # against HBM - algorithmic bytes (bytes_in + bytes_out of the dense boundary) over the event-timed call
            valu = {k: roof[k] for k in ("achieved", "peak", "unit", "frac", "flops", "executed_flops_per_launch", "executed_model",
                                         "frac_necessary", "necessary_flops_per_launch", "frac_algorithmic", "algorithmic_flops_per_launch")}
            valu["bound"] = roof["bound"]
            gbs = alg_bytes / (fwd_ms * 1e-3) / 1e9
            roof.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "bytes": "algorithmic (SURVEY 8d: the dense tensors in, x / z / s / y out)",
                         "achieved_measured_traffic": (traffic / (fwd_ms * 1e-3) / 1e9) if traffic else None,
                         "valu": valu})
            for k in ("flops", "executed_flops_per_launch", "executed_model", "frac_necessary", "necessary_flops_per_launch", "necessary_model",
                      "frac_algorithmic", "algorithmic_flops_per_launch", "note_algorithmic"):
                roof.pop(k, None)
            if cj and cj.get("mfma_f64_ops") is not None:
                roof["mfma_f64_ops_per_launch"] = cj["mfma_f64_ops"]
        if primal and not dense_pile:
            ph = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_config5_primal_phases.txt")), reverse=True)
            if ph:
                roof["phases_source"] = os.path.relpath(ph[0], ROOT) + " (in-kernel cycle counters per phase: make primalprof)"
        if cj:
            waves = B if primal else (B + 3) // 4
            useful = fl_exec / 2.0 / waves / 64.0                               # wave-wide FMA instructions' worth of executed FLOPs
            roof.update({"valu_active": cj.get("valu_active"), "wait_frac": cj.get("wait_frac"),
                         "valu_insts_per_wave": cj.get("valu_insts_per_wave"),
                         "insts_per_useful_fma": (cj["valu_insts_per_wave"] / useful) if cj.get("valu_insts_per_wave") else None,
                         "counters_source": cj["source"]})
        if tj:
            roof["traffic_source"] = tj["source"]
        if rj:
            roof["regs"] = rj
        # (the quoted figures above - traffic, issue counters, registers - come from committed rocprofv3 / compiler runs: stale when the kernel
        #  sources have changed since those files were stamped; ONE flag for the object, the per-file notes say which)
        stale = {n: q.get("stale_note") for n, q in (("traffic", tj), ("counters", cj), ("regs", rj)) if q and q.get("counters_stale")}
        roof["counters_stale"] = bool(stale)
        if stale:
            roof["counters_stale_files"] = stale
        if not a.fwd_only:
            dense_bwd = a.bwd == "dense"
            bkey = key + ("_bwd" if dense_bwd else "_bwd_physical")
            btj = _quoted("traffic", bkey)
            btraffic = (2 * btj["fetch_kb"] + btj["write_kb"]) * 1024.0 if btj else None
            # what the backward must move: G, A, the cotangent and the fp64 iterate in, the seven dense gradients out (SURVEY 8d's
            # figure also counts Q and F as reads: lcp.py:37-64 does not touch them and neither does the kernel)
            balg = ((4 * (m * nz + e * nz + nz) + 8 * (nz + e + 2 * m) + 4 * (nz * nz + nz + m * nz + m + e * nz + e + m * m)) if dense_bwd
                    else 4 * (14 * nb + 7 * nc + 3 * nb) + 4 * (11 * nb + 6 * nc)) * B
            if dense_bwd:
                used = btraffic if btraffic else balg
                roof["bwd"] = {"bound": "hbm",
                               "kernel": ("lcp_big_kernel<64, true, DENSE> / lcp_primal_kernel<..., BWD, DENSE> (lcp.py:37-64 at nineq 256: the seven dense "
                                          "gradients, 302 KB per scene)") if dense_pile else
                                         "lcp_bwd_quad<float,double,%s> (lcp.py:37-64: one factorisation, 1 + 1 KKT solves, the seven dense gradients)"
                                         % ("true" if body_space else "false"),
                               "achieved": used / (bwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": used / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "bytes": "measured traffic" if btraffic else "algorithmic",
                               "traffic": btraffic, "algorithmic_bytes_per_launch": balg, "kernel_ms": bwd_ms,
                               "executed_flops_per_launch": flops.flops_backward_executed_body_space(nz, nc, e) * B if (body_space or dense_pile) else None}
            else:
                # the physical backward moves ~1 KB per scene: it is bound by its one factorisation + 1 + 1 KKT solves (fp64 VALU)
                bfl = (flops.flops_forward_executed_primal(nz, nc, e, 0, True) + 2 * (nc * 102 + 2 * (nz - e) ** 2) if primal
                       else flops.flops_backward_executed_body_space(nz, nc, e) - (2 * nz * nz + 3 * m * nz + m * m + 3 * e * nz)) * B
                roof["bwd"] = {"bound": "valu_fp64",
                               "kernel": ("lcp_primal_kernel<30, true, ...> behind lcp_step_backward_f32" if primal else "lcp_bwd_step_quad")
                                         + " (lcp.py:37-64 contracted through the assembly: gradients w.r.t. the physical inputs)",
                               "achieved": tf(bfl, bwd_ms), "peak": peak, "unit": "TFLOP/s", "frac": tf(bfl, bwd_ms) / peak,
                               "flops": "executed (one formation + LU at the best iterate, 1 + 1 KKT solves)",
                               "traffic": btraffic, "algorithmic_bytes_per_launch": balg, "kernel_ms": bwd_ms,
                               "executed_flops_per_launch": bfl}
            if btj:
                roof["bwd"]["traffic_source"] = btj["source"]
                roof["bwd"]["counters_stale"] = bool(btj.get("counters_stale"))
        shape = ("ten-box pile (4-3-2-1 pyramid), %d pts/interface" % a.pts) if self.pile else "%d-box stack, %d pts/interface" % (a.nbox, a.pts)
        return {
            "config": {"workload": "%s: batch=%d x %d contacts (%s; nz %d, nineq %d, "
                                   "neq %d) per GPU, fp32 I/O, LCP %s, mode=%s, bwd=%s%s"
                                   % (cfg, B, nc, shape, nz, m, e, what, a.mode, "none" if a.fwd_only else a.bwd,
                                      "; step outputs: new_v and the moved pose (engines.py:76-77, bodies.py:80-82), multipliers kept in the "
                                      "workspace in fp64" if a.mode == "fused" else ""),
                       "launch": ("one HIP graph replay per step (forward + backward captured once)" if self.graph is not None
                                  else "eager: one ctypes call per kernel launch"),
                       "device_spin_up_s": round(self.spinup_s, 3), "device_spin_up_steps": self.spinup_steps,
                       "mean_pdipm_iters": float(iters.mean()), "nonzero_status": int((st != 0).sum()),
                       "status_bits": {name: int(((st & bit) != 0).sum()) for name, bit in
                                       (("singular_Q", 1), ("singular_S11", 2), ("singular_T", 4), ("nan", 8), ("truncated", 16))}},
            "roofline": roof,
        }

    def host_side_checks(self):
        """cpu_baseline, cpu_reference and the parity object (512 scenes sampled over the batch)."""
        from lcp_physics_amd.physics import assemble_contacts, fused_step
        from oracle import pdipm_oracle as O
        from tests import parity
        a, B, nz = self.args, self.B, self.nz
        quote = cpu_reference_quote(self.nc)
        out = {"cpu_baseline": cpu_baseline(self.sc_cpu, self.cot_cpu, a.cpu_budget, a.fwd_only),
               "cpu_reference": cpu_reference_live(self.nc, quote) or quote}
        n = min(512, B)
        idx = torch.arange(0, B, max(1, B // n))[:n]
        di = idx.to(self.dev)
        take = lambda t: None if t is None else t[di].cpu()
        sub_cpu = self.sc_cpu.slice(0, B)
        for fl in ("p", "v", "Mdiag", "f", "rest", "fric", "c_n", "c_p1", "c_p2", "c_i1", "c_i2", "Je"):
            setattr(sub_cpu, fl, getattr(self.sc_cpu, fl)[idx])
        kw = {"phys": sub_cpu.phys_dict(), "dt": sub_cpu.dt}
        if a.mode == "dense":
            x_gpu, z_gpu, s_gpu, it_gpu = self.sol.x, self.sol.z, self.sol.s, self.sol.iters
        else:
            # the same call with the multipliers written out (the timed one keeps them in the workspace only)
            chk = fused_step(self.sc, compute=a.compute)
            torch.cuda.synchronize()
            same = bool(torch.equal(chk["v_new"], self.step_out["v_new"]) and torch.equal(chk["iters"], self.step_out["iters"])
                        and torch.equal(chk["p_new"], self.step_out["p_new"]))
            if not same:
                raise SystemExit("bench.py: the step with multipliers differs from the timed step")
            x_gpu, z_gpu, s_gpu, it_gpu = -self.step_out["v_new"].reshape(B, nz), chk["z"], chk["s"], self.step_out["iters"]
        if not a.fwd_only:
            if a.bwd == "physical":
                kw["phys_grads"] = {k: take(v) for k, v in self.pgrads.items()}
            else:
                kw["grads"] = {k: take(t) for k, t in zip("QpGhAbF", self.grads)}
        # identical inputs: the fp32 LCP data the HIP assembly kernel produces for these scenes (all assembling kernels share one
        # contraction-free builder), solved by the oracle in fp64
        lcp_s = assemble_contacts(sub_cpu.to(device=self.dev)) if self.lcp is None else [None if t is None else t[di] for t in self.lcp]
        lcp64 = [None if t is None else t.double().cpu() for t in lcp_s]
        rep, _ = parity.headline_report(O, lcp64, take(x_gpu), take(z_gpu), take(s_gpu), take(it_gpu),
                                        cot=None if a.fwd_only else self.cot_cpu[idx], oracle_stability="always", **kw)
        rep.pop("index_set_mismatch_rows", None) if not rep.get("index_set_mismatches_unmasked") else None
        rep["ref_fp32_vs_fp64_note"] = ("SURVEY 8(d)'s companion: the reference algorithm (the oracle) solved in fp32 arithmetic against itself in fp64 on the same "
                                        "sampled scenes - err_x and the index-set rows it decides differently; index_set_mismatches_oracle_stable = rows where the "
                                        "kernel differs from the fp64 oracle although the oracle keeps its decision in fp32 AND without pivoting (0 = every "
                                        "differing row is one the reference does not decide reproducibly itself)")
        rep["sample"] = "every %d-th scene of the batch" % max(1, B // n)
        # ... and the oracle's OWN assembly of the same scenes (engines.py:31-32, 50-74 restated; fp32 like the kernel's): the LCP
        # data compared entry by entry, and the kernel's new_v against the oracle's solve of what the oracle assembled
        lcp_o = O.assemble_lcp(*sub_cpu.assembly_args())
        worst = 0.0
        for t_hip, t_o in zip(lcp_s, lcp_o):
            if t_o is not None:
                worst = max(worst, float((t_hip.cpu().double() - t_o.double()).abs().max() / max(1e-30, float(t_o.abs().max()))))
        lo64 = [None if t is None else t.double() for t in lcp_o]
        ref_o = O.lcp_forward(*lo64)
        rep["assembly_max_rel_diff_vs_oracle_assembly"] = worst
        rep["fwd_err_x_max_oracle_assembled"] = float(parity.err_x(take(x_gpu).double(), ref_o.x, lo64[0], lo64[1]).max())
        if a.mode == "fused":
            rep["multipliers"] = ("z, s of a repeat of the timed call that writes them out (the timed step returns new_v and the pose "
                                  "only, as the reference's step does); new_v, pose and iteration counts of the two calls: bitwise equal")
        out["parity"] = rep
        return out


HipStackWorkload = HipWorkload          # (the name rounds 1-3 used)


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
    run_rank(args, HipWorkload)


if __name__ == "__main__":
    main()

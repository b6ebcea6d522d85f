"""Time the UNMODIFIED reference solver (lcp/solvers/pdipm.py through oracle/ref_shim.py) on the headline workload's
scenes, on the CPU cores of the machine this runs on.  TEST / MEASUREMENT INFRASTRUCTURE ONLY; needs /root/reference, so it
cannot run on the GPU box: the number it prints is recorded in DESIGN.md §7 and profiles/r01_reference_cpu_timing.json
next to bench.py's `cpu_baseline` (the vectorised port, which does run there).

    PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference.py [--batch 256] [--dtype float64]

The reference is called exactly as `physics/engines.py:76` calls it, but with the whole batch at once (its batched form:
one `LCPFunction(max_iter=10)(Q, p, G, h, A, b, F)` + `.backward`).  Batch-global termination (pdipm.py:116-133) makes
the batched call do at least the work of the slowest scene.
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import pdipm_oracle as O, ref_shim  # noqa: E402


def time_reference(batch=256, dtype="float64", reps=2, pile=False, threads=None, machine="build container (not the GPU box)"):
    """One timing of the unmodified reference on the first `batch` scenes of the workload; `threads`: torch intra-op threads
    (None: the process default).  Returns the record `profiles/r*_reference_cpu_timing.json` keeps."""
    from lcp_physics_amd import scenes
    ref_shim.load_reference()
    dt = getattr(torch, dtype)
    # (torch.set_num_threads breaks MKL's batched 256 x 256 getrf - see bench.py cpu_baseline: the piles keep the process default)
    if threads is not None and not pile:
        torch.set_num_threads(int(threads))
    threads = torch.get_num_threads()
    sc = (scenes.make_pile_scenes(B=batch, seed=5, dtype=torch.float32) if pile else
          scenes.make_stack_scenes(B=batch, nbox=4, pts_per_interface=4, seed=1236, dtype=torch.float32))
    lcp = [None if t is None else t.to(dt) for t in O.assemble_lcp(*sc.assembly_args())]
    cot = torch.randn(batch, lcp[0].shape[1], generator=torch.Generator().manual_seed(4321), dtype=dt)
    best = None
    for _ in range(reps):
        ins = [t.clone().requires_grad_(True) for t in lcp]
        t0 = time.perf_counter()
        x = ref_shim.RefLCPFunction(max_iter=10)(*ins)
        t1 = time.perf_counter()
        x.backward(cot)
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1)
    return {"what": "unmodified reference pdipm (through oracle/ref_shim.py), fwd+bwd, batched call",
            "workload": ("first %d scenes of the configs[4] workload (ten-box pile, 64 contacts, nineq 256)" if pile else
                         "first %d scenes of the headline workload (4-box stack, 16 contacts, nineq 64)") % batch,
            "nc": sc.nc, "batch": batch,
            "dtype": dtype, "cpu_threads": threads, "host_cores": os.cpu_count(), "value": batch / best[0], "unit": "sim steps/s",
            "fwd_s": best[1], "bwd_s": best[2], "machine": machine}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="float64", choices=["float32", "float64"])
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--pile", action="store_true", help="BASELINE configs[4]: the ten-box piles (64 contacts, nineq 256)")
    ap.add_argument("--threads", default=None, help="torch intra-op threads; a comma-separated list sweeps them (one JSON line each, then the best)")
    ap.add_argument("--machine", default="build container (not the GPU box)")
    args = ap.parse_args()
    sweep = [None] if args.threads is None else [int(t) for t in args.threads.split(",")]
    runs = [time_reference(args.batch, args.dtype, args.reps, args.pile, t, args.machine) for t in sweep]
    for r in runs:
        print(json.dumps(r), flush=True)
    if len(runs) > 1:
        best = max(runs, key=lambda r: r["value"])
        print(json.dumps(dict(best, best_of_thread_sweep=[r["cpu_threads"] for r in runs])), flush=True)


if __name__ == "__main__":
    main()

"""CPU, world_size 2 over gloo: the multi-GPU path shards scenes with no data-path collective; only
the barrier / MAX-timing reduction / optional gather use torch.distributed.  The second half runs bench.py's own
rank protocol (`bench.run_rank`: launcher checks, barriers, MAX-over-ranks wall time, n_gpus accounting, the JSON
line) on two gloo ranks with a stand-in workload whose step launches nothing - the HIP launch is what is mocked, not
the protocol - and checks that `bench.py --gpus N` refuses to run when N devices are not there."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_range_partitions_exactly():
    from lcp_physics_amd.shard import shard_range
    for total in (1, 7, 4096, 32768, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(32768, 3, 8) == (3 * 4096, 4 * 4096)       # config 4: 8 x 4096


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from lcp_physics_amd import scenes, shard
    from oracle import pdipm_oracle as O
    r, lr, w = shard.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    full = scenes.make_stack_scenes(B=total, nbox=2, pts_per_interface=2, seed=5, dtype=torch.float64)
    lo, hi = shard.shard_range(total, rank, world)
    mine = full.slice(lo, hi)
    new_v, _, _ = O.solve_dynamics(*mine.assembly_args())       # stands in for the per-rank HIP launch
    shard.barrier()
    t = shard.max_over_ranks(1.0 + rank)
    n = shard.sum_over_ranks(hi - lo)
    allv = shard.gather_scenes(new_v, total)
    if rank == 0:
        ref, _, _ = O.solve_dynamics(*full.assembly_args())
        q.put((t, n, bool(torch.allclose(allv, ref, atol=1e-12)), tuple(allv.shape)))
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_step_equals_single_process():
    world, total = 2, 7                       # uneven split on purpose
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    t, n, same, shape = q.get()
    assert t == 2.0 and n == total and same and shape == (total, 3, 3)


# ---------------------------------------------------------------------------------------------- bench.py's rank protocol
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _NoLaunchWork:
    """Stand-in for bench.HipStackWorkload: same interface, the step is a host-side no-op of known duration."""

    def __init__(self, args, rank, dev):
        self.units_per_step = args.batch
        self.rank, self.calls = rank, 0

    def new_events(self):
        return None

    def step(self, ev=None):
        import time
        self.calls += 1
        time.sleep(0.02 * (1 + self.rank))           # rank 1 is the slow one: the MAX rule must pick it up (long enough for a loaded host)

    def metric_name(self):
        return "protocol test"

    def report(self, events, world):
        return {"config": {"workload": "no-launch stand-in", "calls": self.calls}, "roofline": None}

    def host_side_checks(self):
        return {}


def _bench_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse(["--gpus", str(world), "--steps", "5", "--warmup", "2", "--batch", "4096", "--sustain", "0"])
    out = bench.run_rank(args, _NoLaunchWork, device="cpu")
    if rank == 0:
        q.put(out)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_rank_protocol_gloo_ranks(world):
    """bench.py's rank protocol at world size 2 and at the EIGHT ranks the driver's scaling run starts (the first real
    8-GPU run must not be the first time eight ranks ever start): rendezvous on 127.0.0.1, barriers, MAX over ranks, per-rank records."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    out = q.get()
    slow = 20.0 * world                                     # the last rank sleeps 20 ms x world per step
    assert out["n_gpus"] == world and out["steps"] == 5 and out["warmup"] == 2
    assert out["config"]["global_batch"] == 4096 * world and out["scaling"] == "weak"
    assert out["config"]["calls"] == 7 + 3                  # warm-up + timed steps, then the three steps of the host-issue probe: nothing else
    assert out["ms_per_step"] >= slow                       # MAX over ranks: the slowest rank's time per step
    assert abs(out["value"] - 4096 * world * 5 / (out["ms_per_step"] * 5e-3)) < 1e-6 * out["value"]
    assert out["cpu_baseline"] is None                      # reported at N = 1 only
    # every rank's own clock is in the line (the metric takes the MAX): rank r sleeps (r + 1) x 20 ms per step
    pr = out["per_rank"]
    assert [r["rank"] for r in pr] == list(range(world))
    assert pr[-1]["ms_per_step"] >= slow > pr[0]["ms_per_step"] >= 20.0
    assert abs(max(r["ms_per_step"] for r in pr) - out["ms_per_step"]) < 0.25 * slow
    # the host-issue probe: per rank, and the ceiling it implies (the stand-in's "issue" is its sleep)
    assert all(r["host_us_per_step"] >= 20e3 * (r["rank"] + 1) for r in pr)
    assert out["host_us_per_step"] == max(r["host_us_per_step"] for r in pr)
    assert abs(out["host_bound_ceiling"]["value"] - 4096 * world / (out["host_us_per_step"] * 1e-6)) < 1e-6 * out["host_bound_ceiling"]["value"]


def _run_bench(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                          env=env, timeout=timeout)


def test_bench_refuses_more_ranks_than_devices():
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run_bench(["--gpus", str(n + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing to run" in (r.stderr + r.stdout) and "n_gpus" not in r.stdout


def test_bench_rejects_gpus_flag_that_disagrees_with_the_launcher():
    r = _run_bench(["--gpus", "4", "--steps", "1", "--warmup", "0"],
                   env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)

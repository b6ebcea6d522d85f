"""GPU: the HIP step under `lcp_physics_amd.shard` and bench.py's multi-rank launcher on however many
devices the box has.  Scenes never read each other, so a shard's results must be BITWISE the corresponding slice of the
unsharded launch, whatever the shard boundaries are."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_hip_step_is_bitwise_the_unsharded_step(world):
    from lcp_physics_amd import scenes, shard
    from lcp_physics_amd.physics import fused_step
    total = 203                                                     # uneven on purpose (not a multiple of 4 scenes per wave)
    ndev = torch.cuda.device_count()
    full_cpu = scenes.make_stack_scenes(B=total, nbox=4, pts_per_interface=4, seed=77, dtype=torch.float32)
    full = fused_step(full_cpu.to(device="cuda:0"))
    torch.cuda.synchronize()
    keys = ("v_new", "p_new", "z", "s", "y", "iters", "status")
    for rank in range(world):
        lo, hi = shard.shard_range(total, rank, world)
        dev = "cuda:%d" % (rank % ndev)                              # every device the box has takes its share of the ranks
        out = fused_step(full_cpu.slice(lo, hi).to(device=dev))
        torch.cuda.synchronize(dev)
        for k in keys:
            assert torch.equal(out[k].cpu(), full[k][lo:hi].cpu()), (world, rank, k)


def _bench(argv, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                          env=env, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


SMALL = ["--steps", "3", "--warmup", "1", "--batch", "256", "--no-cpu-baseline", "--sustain", "0"]


def test_bench_gpus_flag_launches_that_many_ranks_or_refuses():
    ndev = torch.cuda.device_count()
    r = _bench(["--gpus", str(ndev + 1)] + SMALL)                      # one more than there are: a clear refusal, no JSON line
    assert r.returncode != 0 and "refusing to run" in (r.stderr + r.stdout) and "n_gpus" not in r.stdout
    if ndev >= 2:
        r = _bench(["--gpus", "2"] + SMALL)
        assert r.returncode == 0, r.stderr[-2000:]
        j = _json_line(r.stdout)
        assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 512 and "devices_used" not in j


def test_bench_two_ranks_run_the_hip_path_through_the_launcher():
    """The self-launch path end to end with the real workload: two ranks (sharing the device when the box has one - the
    testing aid, gloo for the barriers), each with its own scenes."""
    ndev = torch.cuda.device_count()
    extra = [] if ndev >= 2 else ["--share-devices"]
    r = _bench(["--gpus", "2"] + extra + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 512
    assert j.get("devices_used", 2) == min(ndev, 2)
    assert j["roofline"]["fwd_ms"] > 0 and j["value"] > 0
    assert sorted(r["rank"] for r in j["per_rank"]) == [0, 1] and all(r["ms_per_step"] > 0 for r in j["per_rank"])   # auditable rank by rank


def test_bench_single_rank_line_has_the_contract_fields():
    r = _bench(SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    rf = j["roofline"]
    assert j["n_gpus"] == 1 and rf["bound"] == "valu_fp64"
    # `frac` = the FLOPs the kernel executes over the FP64 vector peak (an efficiency: below 1); `frac_algorithmic` = SURVEY 8d's count
    # of the reference's dense formulation over the same time (an algorithmic saving: may exceed 1)
    assert rf["flops"] == "executed" and 0 < rf["frac"] < 1.0 and rf["frac"] < rf["frac_algorithmic"]
    assert rf["bwd"]["bound"] == "hbm" and 0 < rf["bwd"]["frac"] < 1.0
    assert [r["rank"] for r in j["per_rank"]] == [0] and j["per_rank"][0]["fwd_ms"] > 0

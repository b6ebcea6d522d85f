"""Multi-GPU sharding of independent scenes.

Scenes never read each other, so the batch shards embarrassingly: one process per GPU, rank r of
N owns a contiguous slice of the scene indices, and there is NO collective on the solve path.
`torch.distributed` (RCCL on GPUs, gloo in the CPU tests) is only used for the barrier and the
max-over-ranks timing reduction of the benchmark, and for an optional result gather.
"""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of `total` scenes owned by `rank` (remainder spread from rank 0)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def under_launcher():
    """True when this process was started by torchrun / torch.distributed.run (or by `launch_ranks`)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, script, argv, share_devices=False, device_count=None, timeout=None):
    """Start `n` ranks of `script argv...`, one per visible GPU, under torch.distributed.run on this node
    (rendezvous on 127.0.0.1) and return the launcher's exit code.  Refuses (RuntimeError) when fewer than
    `n` devices are visible - a rank is never folded onto somebody else's GPU unless `share_devices` (a
    testing aid: several ranks on one device, gloo instead of RCCL) is set.  HIP_VISIBLE_DEVICES is inherited,
    so the ranks number the devices the parent sees."""
    if n < 1:
        raise RuntimeError("launch_ranks: need at least one rank (got %d)" % n)
    if device_count is None:
        device_count = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not share_devices and device_count < n:
        raise RuntimeError("%d ranks requested but only %d GPU(s) visible on this node: refusing to run "
                           "(one process per GPU, no device sharing)" % (n, device_count))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env, timeout=timeout)


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def rank_device(local_rank, share_devices=False):
    """The GPU of this rank: device `local_rank`, and an error - not a modulo - when it does not exist."""
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n == 0:
        raise RuntimeError("no GPU visible: the HIP path has no CPU fallback")
    if local_rank >= n:
        if not share_devices:
            raise RuntimeError("local rank %d has no GPU of its own (%d visible)" % (local_rank, n))
        local_rank %= n
    return torch.device("cuda", local_rank)


def reduce_device(device):
    """Where the small timing reductions live: on the GPU under RCCL, on the host under gloo."""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        return "cpu"
    return device


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX-reduce a python float over all ranks (the benchmark's wall-time rule)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_scenes(local, total, device=None):
    """Optional: all-gather per-rank results [B_local, ...] into [total, ...] (not on the solve path).
    Shards may differ by one scene, so they are padded to the largest shard."""
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()
    sizes = [shard_range(total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(outs, sizes)], 0)

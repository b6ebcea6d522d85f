"""Multi-GPU sharding of independent scenes.

Scenes never read each other, so the batch shards embarrassingly: one process per GPU, rank r of
N owns a contiguous slice of the scene indices, and there is NO collective on the solve path.
`torch.distributed` (RCCL on GPUs, gloo in the CPU tests) is only used for the barrier and the
max-over-ranks timing reduction of the benchmark, and for an optional result gather.
"""
import os

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


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


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

"""Multi-GPU sharding of independent scenes.

Scenes never read each other, so the batch shards embarrassingly: one process per GPU, rank r of
N owns a contiguous slice of the scene indices, and there is NO collective on the solve path.
`torch.distributed` is only used for the barrier and the max-over-ranks timing reduction of the benchmark, and for an
optional result gather: RCCL on GPUs when it comes up on every rank, gloo otherwise and in the CPU tests (`control_plane()`).
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


# The control plane of a multi-rank run.  There is no collective on the solve path, so nothing about a run's RESULT depends on which
# library carries the barriers and the three timing scalars: gloo (TCP on 127.0.0.1) is brought up first and always works; RCCL
# ("nccl" on ROCm) is then TRIED - communicator + a one-element all_reduce under a blocking-wait timeout - and used for the barriers /
# reductions only if every rank succeeded (agreed over gloo).  A node where RCCL cannot come up (IPC mode, a missing xGMI link, two ranks
# on one device) still produces its line, with `control_plane()` saying why.
_CTRL = {"group": None, "backend": None, "note": None}


def control_plane():
    """What carries the barriers / timing reductions of this run, for the benchmark line: e.g. "rccl", "gloo (rccl init failed: ...)"."""
    if not dist.is_initialized():
        return "single process"
    if _CTRL["backend"] == "nccl":
        return "rccl"
    return "gloo" + ((" (%s)" % _CTRL["note"]) if _CTRL["note"] else "")


def _try_rccl(local_rank, world, timeout_s):
    """Collective over the default (gloo) group.  Returns (ok on EVERY rank, note)."""
    import datetime
    ok, note, group = 1.0, None, None
    try:
        n = torch.cuda.device_count()
        dev = torch.device("cuda", local_rank % max(n, 1))
        torch.cuda.set_device(dev)
        group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
        t = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(t, group=group)
        torch.cuda.synchronize(dev)
        if int(round(float(t.item()))) != world:
            raise RuntimeError("all_reduce returned %r for %d ranks" % (float(t.item()), world))
    except Exception as ex:                                   # noqa: BLE001 - any failure of the optional transport means "use gloo"
        ok, note = 0.0, "rccl init failed: %s" % (str(ex).strip().splitlines() or [type(ex).__name__])[0][:160]
    flag = torch.tensor([ok], dtype=torch.float64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)               # (default group: gloo)
    if float(flag.item()) < 0.5:
        return False, note or "rccl init failed on another rank", None
    return True, None, group


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process).  `backend=None`: gloo as the
    control plane, then RCCL if it comes up on every rank (see `_CTRL` above); "gloo": gloo only (the CPU tests, shared devices)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        want_rccl = backend in (None, "nccl") and torch.cuda.is_available()
        if want_rccl:
            os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")      # a collective that cannot complete raises after the timeout instead of hanging
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        _CTRL.update(group=None, backend="gloo", note=None)
        if want_rccl:
            ok, note, group = _try_rccl(local_rank, world, int(os.environ.get("LCP_RCCL_TIMEOUT_S", "60")))
            if ok:
                _CTRL.update(group=group, backend="nccl", note=None)
            else:
                _CTRL.update(note=note)
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
    if dist.is_initialized() and _CTRL["backend"] != "nccl":
        return "cpu"
    return device


def barrier():
    if dist.is_initialized():
        dist.barrier(group=_CTRL["group"])


def max_over_ranks(value, device="cpu"):
    """MAX-reduce a python float over all ranks (the benchmark's wall-time rule)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_CTRL["group"])
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_CTRL["group"])
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
    on_gpu = pad.is_cuda
    if on_gpu and _CTRL["backend"] != "nccl":                   # gloo moves host memory
        pad = pad.cpu()
        outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=_CTRL["group"] if pad.is_cuda else None)
    if on_gpu and not pad.is_cuda:
        outs = [o.to(local.device) for o in outs]
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(outs, sizes)], 0)

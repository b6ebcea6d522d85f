"""CPU, world_size 2 over gloo: the multi-GPU path shards scenes with no data-path collective; only
the barrier / MAX-timing reduction / optional gather use torch.distributed."""
import os
import socket

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

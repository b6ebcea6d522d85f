"""Upper bound of what pivoting in 3 x 3 body blocks could buy the config-5 forward, measured instead of argued:
library variants of lcp_primal_pin.hip built with -DLCP_PRIMAL_EXP_PREP_EVERY=N run the per-pivot chain of the lane-grid LU (reciprocal,
multipliers, lane masks, hand-over across the DPP rows) for every N-th pivot only and the multiply-adds of all thirty - the factors are
WRONG, the instruction stream is the one a block-pivot LU could not beat (it would still pay a closed-form 3 x 3 inverse and the block's
multipliers per block).  Every scene is forced through all ten iterations (not_improved_lim = 1000), so the variants execute the same
number of factorisations whatever their iterates look like.
    LCP_HIP_LIB=.../variants/<name>.so python tools/experiments/block_pivot_bound.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lcp_physics_amd import scenes
from lcp_physics_amd.physics.batched_world import solve_dynamics
from lcp_physics_amd.physics.contacts import ContactBuffers
B = 4096
sc = scenes.make_pile_scenes(B=B, seed=5, dtype=torch.float32).to('cuda')
cb = ContactBuffers(B, sc.nb, sc.nc, 'cuda')
cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2
count = torch.full((B,), sc.nc, dtype=torch.int32, device='cuda')
run = lambda out=None: solve_dynamics(B, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.f, sc.rest, sc.fric, cb, sc.Je, sc.dt,
                                      ws=None if out is None else out["ws"], out=out, pinned=True, not_improved_lim=1000)
out = run(); torch.cuda.synchronize()
for _ in range(20): out = run(out)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(100): out = run(out)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 100
print("%-14s forward %.4f ms per launch of %d scenes (%.2f M sim steps/s), mean iterations %.2f, NaN scenes %d" % (
    os.path.basename(os.environ.get("LCP_HIP_LIB", "default")), ms, B, B / ms / 1e3, float(out["iters"].float().mean()), int((out["status"] & 8 != 0).sum())))

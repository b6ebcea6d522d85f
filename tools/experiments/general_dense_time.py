"""Timing of the dense boundary on batches the four-scenes-per-wave kernels do not take (A/B aid for the wave-per-scene kernels):
random general LCPs (class 0) and contact-structured LCPs with a dense SPD Q (class 1).  python tools/experiments/general_dense_time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lcp_physics_amd import scenes
from lcp_physics_amd.lcp import lcp_backward, lcp_solve
from oracle import pdipm_oracle as O
B = 4096
def timeit(lcp, label):
    g = [None if t is None else t.to("cuda").contiguous() for t in lcp]
    sol = lcp_solve(*g); cot = torch.randn(B, g[0].shape[1], device="cuda"); gr = lcp_backward(sol, cot); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(5):
        ev[0].record(); lcp_solve(*g, ws=sol.ws, out=sol); ev[1].record(); lcp_backward(sol, cot, out=gr); ev[2].record(); torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]) / 5; tb += ev[1].elapsed_time(ev[2]) / 5
    print("%-44s forward %.3f ms  backward %.3f ms  (%d scenes, mean iterations %.1f)" % (label, tf, tb, B, float(sol.iters.float().mean())))
timeit(scenes.make_random_lcp(B, 15, 64, 3, seed=1, dtype=torch.float32), "general LCPs nz 15, nineq 64, neq 3")
sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=31, dtype=torch.float32)
lcp = [t.clone() if t is not None else None for t in O.assemble_lcp(*sc.assembly_args())]
g = torch.Generator().manual_seed(3); L = torch.randn(B, 15, 15, generator=g)
lcp[0] = lcp[0] + 0.05 * (L @ L.transpose(1, 2)) * lcp[0].diagonal(dim1=1, dim2=2).min(dim=1)[0].reshape(B, 1, 1)
timeit(lcp, "contact-structured, dense SPD Q (class 1)")

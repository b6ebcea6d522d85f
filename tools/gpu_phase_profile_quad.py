"""Per-phase cycle breakdown of the four-scenes-per-wave forward kernel of the contact-list entry points (the kernel bench.py
times).  Needs the -DLCP_Q_PROFILE build:
    make -C lcp_physics_amd/csrc quadprof
    LCP_HIP_LIB=tools/liblcp_quadprof.so python tools/gpu_phase_profile_quad.py [B] [nbox]
(the profiling build writes the cycle record of a wave over the tail of its first scene's `s` output)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lcp_physics_amd import scenes
from lcp_physics_amd.physics import fused_step
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nbox = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=4, seed=1236, dtype=torch.float32).to(device="cuda")
out = None
for rep in range(3):
    out = fused_step(sc, out=out, ws=None if out is None else out["ws"])
    torch.cuda.synchronize()
t = out["s"][::4, -13:].double().cpu()           # one record per wave (lane 0 = first scene of the wave)
names = ["residuals + d", "factor: formation", "factor: LU", "solve_kkt: products before", "solve_kkt: triangular sweeps",
         "solve_kkt: products after", "bookkeeping / best iterate", "step lengths, sigma, update"]
tot = t[:, :8].sum(1)
print("B=%d nbox=%d  mean cycles per wave (clock64 ticks): total of the phases %.0f   (iters %.1f)   prologue %.0f   whole kernel (first to last instruction of the wave) %.0f" % (
    B, nbox, tot.mean(), t[:, 8].mean(), t[:, 9].mean(), t[:, 10].mean()))
print("  prologue: kernel start -> every input loaded and assembled %.0f, -> LDS barrier + Jacobian columns to registers %.0f, -> loop %.0f" % (
    t[:, 11].mean(), t[:, 12].mean(), (t[:, 9] - t[:, 11] - t[:, 12]).mean()))
for k, n in enumerate(names):
    print("  %-32s mean %10.0f  (%.1f%%)" % (n, t[:, k].mean(), 100 * t[:, k].mean() / tot.mean()))

"""Per-phase cycle breakdown of lcp_fwd_solo (one scene per wavefront: batches of at most 1024 scenes, BASELINE configs[1]).
Needs the -DLCP_SOLO_PROFILE build:
    make -C lcp_physics_amd/csrc soloprof
    LCP_HIP_LIB=tools/liblcp_soloprof.so python tools/gpu_phase_profile_solo.py [B] [nbox]
(the profiling build writes the cycle record of a wave over the tail of its scene's `s` output)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lcp_physics_amd import scenes
from lcp_physics_amd.physics import fused_step
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nbox = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=4, seed=1236, dtype=torch.float32).to(device="cuda")
out = None
for rep in range(3):
    out = fused_step(sc, out=out, ws=None if out is None else out["ws"], path="solo")
    torch.cuda.synchronize()
t = out["s"][:, -11:].double().cpu()
names = ["residuals + d", "factor: formation", "factor: LU", "solve_kkt: products before", "solve_kkt: triangular sweeps",
         "solve_kkt: products after", "bookkeeping / best iterate", "step lengths, sigma, update", "prologue (loads, assembly)"]
tot = t[:, 10]
print("B=%d nbox=%d  mean cycles per wave (clock64 ticks): kernel %.0f, in the phases %.0f   (iters %.1f)" % (B, nbox, tot.mean(), t[:, :9].sum(1).mean(), t[:, 9].mean()))
for k, n in enumerate(names):
    print("  %-32s mean %10.0f  (%.1f%%)" % (n, t[:, k].mean(), 100 * t[:, k].mean() / tot.mean()))

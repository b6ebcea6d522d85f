"""Per-phase cycle breakdown of the four-scenes-per-wave forward kernel (needs a -DLCP_Q_PROFILE build:
LCP_HIP_LIB=tools/liblcp_prof.so python tools/gpu_phase_profile_quad.py [B])."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lcp_physics_amd import _lib, scenes
from lcp_physics_amd.lcp import lcp_solve
from lcp_physics_amd.physics import assemble_contacts
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=1236, dtype=torch.float32).to(device="cuda")
lcp = assemble_contacts(sc)
lib = _lib.load()
trace = torch.zeros(B, 10, 4, dtype=torch.float64, device="cuda")
for rep in range(2):
    trace.zero_()
    lib.lcp_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
    sol = lcp_solve(*lcp)
    torch.cuda.synchronize()
lib.lcp_debug_set_trace(None)
t = trace.reshape(B, 40)[::4, :9].cpu()          # one record per wave (lane 0 = first scene of the wave)
names = ["residuals + d", "factor: W load + diag", "factor: LU", "solve_kkt: products before", "solve_kkt: triangular sweeps",
         "solve_kkt: products after", "bookkeeping / best iterate", "step lengths, sigma, update"]
tot = t[:, :8].sum(1)
print("B=%d  mean cycles per wave: total %.0f   (iters %.1f)" % (B, tot.mean(), t[:, 8].mean()))
for k, n in enumerate(names):
    print("  %-32s mean %10.0f  (%.1f%%)" % (n, t[:, k].mean(), 100 * t[:, k].mean() / tot.mean()))

import sys, time, os
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcp_physics_amd import scenes
from lcp_physics_amd.physics.batched_world import solve_dynamics, rows_pin_leading_coordinates
from lcp_physics_amd.physics.contacts import ContactBuffers
nbox, pts = int(sys.argv[1]), int(sys.argv[2])
B = 4096
sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=5, dtype=torch.float32).to('cuda')
cb = ContactBuffers(B, sc.nb, sc.nc, 'cuda')
cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2
count = torch.full((B,), sc.nc, dtype=torch.int32, device='cuda')
PINNED = rows_pin_leading_coordinates(sc.Je) and "nohint" not in sys.argv            # LCP_HINT_PINNED, checked once on the host
run = lambda out=None: solve_dynamics(B, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.f, sc.rest, sc.fric, cb, sc.Je, sc.dt, ws=None if out is None else out["ws"], out=out, pinned=PINNED)
out = run(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): out = run(out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print("nb", sc.nb, "nc", sc.nc, "pinned hint", PINNED, "ms/step", dt * 1e3, "steps/s", B / dt)
if "primalprof" in os.environ.get("LCP_HIP_LIB", ""):
    m = 4 * sc.nc
    pc = out["s"][:, m - 8:m - 2].double().mean(dim=0).tolist()
    print("cycles per scene: residuals %.0f  formation %.0f  LU %.0f  bookkeeping %.0f  solve_kkt %.0f  steps + update %.0f  total %.0f" % (*pc, sum(pc)))
if "bigprof" in os.environ.get("LCP_HIP_LIB", ""):
    m = 4 * sc.nc
    pc = out["s"][:, m - 8:m - 1].double().mean(dim=0).tolist()
    print("cycles: residuals %.0f factor %.0f steps %.0f solve_kkt %.0f (sweeps %.0f) | W load %.0f LU %.0f" % tuple(pc))

"""BASELINE configs[4] (4096 scenes x 64 contacts, nineq 256) - the IN-KERNEL PHASE COUNTERS of lcp_primal_kernel and the A/B against the
contact-space kernel.  The bench line of this configuration is `python bench.py --config 4` (roofline, cpu_baseline, parity); this
script only serves the instrumented library:

    make -C lcp_physics_amd/csrc primalprof
    LCP_HIP_LIB=$PWD/tools/liblcp_primalprof.so python tools/config5_phases.py [B]       cycles per phase and scene
    python tools/config5_phases.py [B] big                                             lcp_big.hip (contact space) timed instead
    python tools/config5_phases.py [B] nohint                                          the general form (no LCP_HINT_PINNED)
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcp_physics_amd import _lib, scenes
from lcp_physics_amd.physics.batched_world import solve_dynamics, rows_pin_leading_coordinates
from lcp_physics_amd.physics.contacts import ContactBuffers
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
sc = scenes.make_pile_scenes(B=B, seed=5, dtype=torch.float32).to('cuda')
cb = ContactBuffers(B, sc.nb, sc.nc, 'cuda')
cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2
count = torch.full((B,), sc.nc, dtype=torch.int32, device='cuda')
PINNED = rows_pin_leading_coordinates(sc.Je) and "nohint" not in sys.argv
if "big" in sys.argv:
    _lib.set_path("big")
run = lambda out=None: solve_dynamics(B, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.f, sc.rest, sc.fric, cb, sc.Je, sc.dt,
                                      ws=None if out is None else out["ws"], out=out, pinned=PINNED)
out = run(); torch.cuda.synchronize()
for _ in range(3): out = run(out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): out = run(out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 20
print("config 5 forward (%s%s): %.4f ms per launch of %d scenes = %.2f M sim steps/s, mean iterations %.2f" % (
    "lcp_big.hip, contact space" if "big" in sys.argv else "lcp_primal_kernel", ", LCP_HINT_PINNED" if PINNED else "", dt * 1e3, B, B / dt / 1e6,
    float(out["iters"].float().mean())))
if "prof" in os.path.basename(os.environ.get("LCP_HIP_LIB", "")) and "bigprof" not in os.environ.get("LCP_HIP_LIB", ""):
    pc = out["s"][:, 248:255].double().mean(dim=0).tolist()
    print("cycles per scene: residuals %.0f  formation %.0f  LU %.0f  bookkeeping %.0f  solve_kkt %.0f (of which the triangular sweeps %.0f)  steps + update %.0f  total %.0f"
          % (pc[0], pc[1], pc[2], pc[3], pc[4], pc[6], pc[5], sum(pc[:6])))
if "bigprof" in os.environ.get("LCP_HIP_LIB", ""):
    pc = out["s"][:, 248:255].double().mean(dim=0).tolist()
    print("factor split: W load + diag %.0f   LU loop %.0f" % (pc[5], pc[6]))
    for w in range(4):
        pm = out["z"][:, 232 + 5 * w:237 + 5 * w].double().mean(dim=0).tolist()
        print("blocked LU, wave %d: publish %.0f  barrier %.0f   panel %.0f   barrier %.0f   trailing MFMA %.0f" % (w, pm[4], pm[0], pm[1], pm[2], pm[3]))
    tot = sum(pc[:4])
    print("cycles per scene: residuals %.0f  factor %.0f  steps+bookkeeping %.0f  solve_kkt %.0f (of which triangular sweeps %.0f)  total %.0f" % (pc[0], pc[1], pc[2], pc[3], pc[4], tot))

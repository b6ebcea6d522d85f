"""BASELINE config 5 (4096 scenes x 64 contacts, nineq 256): lcp_solve_dynamics_f32 on the pile scenes, checked against the
generic kernels on a few scenes and timed.   python tools/bench_config5.py [B]"""
import sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcp_physics_amd import _lib, flops, scenes
from lcp_physics_amd.physics.batched_world import solve_dynamics, rows_pin_leading_coordinates
from lcp_physics_amd.physics.contacts import ContactBuffers
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = scenes.make_pile_scenes(B=B, seed=5, dtype=torch.float32).to('cuda')
cb = ContactBuffers(B, sc.nb, sc.nc, 'cuda')
cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2
count = torch.full((B,), sc.nc, dtype=torch.int32, device='cuda')
# LCP_HINT_PINNED: checked once on the host, as ContactWorld / fused_step do (argument "nohint": the general form, for the A/B)
PINNED = rows_pin_leading_coordinates(sc.Je) and "nohint" not in sys.argv
run = lambda out=None: solve_dynamics(B, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.f, sc.rest, sc.fric, cb, sc.Je, sc.dt,
                                      ws=None if out is None else out["ws"], out=out, pinned=PINNED)
out = run(); torch.cuda.synchronize()
nchk = min(B, 64)
_lib.set_path("generic")
sub = lambda t: t[:nchk].contiguous()
cb2 = ContactBuffers(nchk, sc.nb, sc.nc, 'cuda')
cb2.c_n, cb2.c_p1, cb2.c_p2, cb2.c_i1, cb2.c_i2 = sub(sc.c_n), sub(sc.c_p1), sub(sc.c_p2), sub(sc.c_i1), sub(sc.c_i2)
ref = solve_dynamics(nchk, sc.nb, sc.nc, 3, sub(count), sub(sc.Mdiag), sub(sc.v), sub(sc.f), sub(sc.rest), sub(sc.fric), cb2, sub(sc.Je), sc.dt)
torch.cuda.synchronize()
_lib.set_path("auto")
d = (out["v_new"][:nchk] - ref["v_new"]).abs().max()
print("max |v_new|", float(ref["v_new"].abs().max()), "max |z|", float(ref["z"].abs().max()), "max |z - z_generic|", float((out["z"][:nchk] - ref["z"]).abs().max()),
      "max |s - s_generic|", float((out["s"][:nchk] - ref["s"]).abs().max()))
print("nb", sc.nb, "nc", sc.nc, "max |v_new - generic| on %d scenes: %.3e" % (nchk, float(d)), "iters", float(out["iters"].float().mean()), float(ref["iters"].float().mean()),
      "status!=0", int((out["status"] != 0).sum()))
if len(sys.argv) > 2 and sys.argv[2] == "big":      # contact-space lcp_big.hip instead of the body-space lcp_primal.hip (A/B)
    _lib.set_path("big")
    out = run(); torch.cuda.synchronize()
for _ in range(3): out = run(out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): out = run(out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
import json
from lcp_physics_amd.physics.batched_world import fused_step_backward
cot = torch.randn(B, sc.nb, 3, device='cuda')
pg = fused_step_backward(sc, out, cot); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): pg = fused_step_backward(sc, out, cot)
torch.cuda.synchronize()
dtb = (time.perf_counter() - t) / 10
if len(sys.argv) > 2 and sys.argv[2] == "dense":
    # the same piles through the dense LCPFunction boundary (lcp_pdipm_forward_f32 / _backward_f32): classification on the
    # device, then lcp_big.hip for the contact-structured scenes
    from lcp_physics_amd.lcp import lcp_backward, lcp_solve
    from lcp_physics_amd.physics import assemble_contacts
    lcp = assemble_contacts(sc)
    sol = lcp_solve(*lcp)
    cot = torch.randn(B, lcp[0].shape[1], device='cuda')
    grads = lcp_backward(sol, cot)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    reps = 3
    tf = tb = 0.0
    for _ in range(reps):
        ev[0].record(); sol = lcp_solve(*lcp, ws=sol.ws, out=sol); ev[1].record(); lcp_backward(sol, cot, out=grads); ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]) / reps; tb += ev[1].elapsed_time(ev[2]) / reps
    dx = float((sol.x + out["v_new"].reshape(B, -1)).abs().max())
    print(json.dumps({"metric": "sim steps/s, BASELINE config 5 through the dense LCPFunction boundary (nz 33, nineq 256, neq 3)",
                      "forward_value": B / (tf * 1e-3), "fwd_bwd_value": B / ((tf + tb) * 1e-3), "unit": "sim steps/s", "batch": B,
                      "fwd_ms": tf, "bwd_ms": tb, "max_abs_x_plus_v_new_of_the_contact_list_entry": dx,
                      "generic_kernels_before": 2.3e3}))
    sys.exit(0)
print(json.dumps({"metric": "sim steps/s, BASELINE config 5 (batch x 64 contacts, nineq 256, nz 33, neq 3), forward (lcp_solve_dynamics_f32)",
                  "value": B / dt, "unit": "sim steps/s", "batch": B, "ms_per_step": dt * 1e3,
                  "backward_ms": dtb * 1e3, "fwd_bwd_value": B / (dt + dtb),
                  "mean_pdipm_iters": float(out["iters"].float().mean()), "max_abs_diff_vs_generic_kernels": float(d),
                  "executed_flops_per_scene": flops.flops_forward_executed_primal(3 * sc.nb, sc.nc, 3, float(out["iters"].float().mean()), PINNED),
                  "frac_of_fp64_vector_peak": flops.flops_forward_executed_primal(3 * sc.nb, sc.nc, 3, float(out["iters"].float().mean()), PINNED) * B / dt / 78.6e12,
                  "algorithmic_flops_per_scene_survey_8d": flops.flops_forward(3 * sc.nb, 4 * sc.nc, 3, float(out["iters"].float().mean())),
                  "kernel": "lcp::big::lcp_big_kernel<64> (contact space, 128 x 128: blocked LU, trailing updates on v_mfma_f64_16x16x4_f64)"
                            if len(sys.argv) > 2 and sys.argv[2] == "big" else
                            ("lcp::primal::lcp_primal_kernel<32, ..., PIN> (body space, the 30 free coordinates' system, one wave per scene; LCP_HINT_PINNED)"
                             if PINNED else "lcp::primal::lcp_primal_kernel<40> (body space, 36 x 36 systems, one wave per scene)")}))
if "primalprof" in os.environ.get("LCP_HIP_LIB", ""):
    pc = out["s"][:, 248:254].double().mean(dim=0).tolist()
    print("cycles per scene: residuals %.0f  formation %.0f  LU %.0f  bookkeeping %.0f  solve_kkt %.0f  steps + update %.0f  total %.0f"
          % (pc[0], pc[1], pc[2], pc[3], pc[4], pc[5], sum(pc)))
import os
if "bigprof" in os.environ.get("LCP_HIP_LIB", ""):
    pc = out["s"][:, 248:255].double().mean(dim=0).tolist()
    print("factor split: W load + diag %.0f   LU loop %.0f" % (pc[5], pc[6]))
    for w in range(4):
        pm = out["z"][:, 232 + 5 * w:237 + 5 * w].double().mean(dim=0).tolist()
        print("blocked LU, wave %d: publish %.0f  barrier %.0f   panel %.0f   barrier %.0f   trailing MFMA %.0f" % (w, pm[4], pm[0], pm[1], pm[2], pm[3]))
    tot = sum(pc[:4])
    print("cycles per scene: residuals %.0f  factor %.0f  steps+bookkeeping %.0f  solve_kkt %.0f (of which triangular sweeps %.0f)  total %.0f" % (pc[0], pc[1], pc[2], pc[3], pc[4], tot))

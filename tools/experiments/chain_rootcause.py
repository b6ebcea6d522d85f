"""Round 6: the five chain cases that go through lcp_primal_kernel<56, ...> (tests/test_hip_primal.py::
test_chains_of_joints_up_to_24_equality_rows), forward only, against the generic kernel - run per library variant
(LCP_HIP_LIB=...) by tools/gpu_calls/r06_*.sh; prints one line per case: worst scaled |v_new - v_new(generic)|, iteration counts."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_hip_primal import _solve, _with_joint_rows          # noqa: E402
from lcp_physics_amd import scenes                                   # noqa: E402
from lcp_physics_amd.physics.batched_world import fused_step         # noqa: E402

CASES = [(8, 2, 16), (12, 2, 16), (8, 2, 24), (10, 2, 20), (11, 2, 20), (4, 4, 12)]
B = 32
worst = 0.0
for nbox, pts, e in CASES:
    sc = _with_joint_rows(scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=900 + nbox + e, dtype=torch.float32), e)
    count = torch.full((B,), sc.nc, dtype=torch.int32)
    scg, out = _solve(sc, count)
    gen = fused_step(scg, path="generic")
    torch.cuda.synchronize()
    va, vb = out["v_new"].double().cpu(), gen["v_new"].double().cpu()
    scale = vb.abs().reshape(B, -1).max(dim=1)[0].clamp_min(1.0)
    err = (va - vb).abs().reshape(B, -1).max(dim=1)[0] / scale
    it_a, it_b = out["iters"].cpu(), gen["iters"].cpu()
    n = 3 * sc.nb + e
    print("case %2d-%d-%2d n=%2d  err max %.3e  bad scenes %2d/%d  iters differ %2d  status|=%d" % (
        nbox, pts, e, n, float(err.max()), int((err > 2e-6).sum()), B, int((it_a != it_b).sum()), int(out["status"].cpu().max())),
        " first bad:", [(int(i), "%.1e" % float(err[i]), int(it_a[i]), int(it_b[i])) for i in torch.nonzero(err > 2e-6).flatten()[:4]])
    if n > 40:
        worst = max(worst, float(err.max()))
print("RESULT", "PASS" if worst <= 2e-6 else "FAIL", "%.3e" % worst)

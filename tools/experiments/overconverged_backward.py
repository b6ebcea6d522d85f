"""Is the NaN of the contact-space backward at an over-converged iterate a property of that backward (any forward that runs a
step further triggers it) ?  Contact-space forward with eps = 0 (never stops on the residual) and more iterations."""
import sys, torch
sys.path.insert(0, '/root/repo')
from lcp_physics_amd import _lib, scenes
from lcp_physics_amd.physics import fused_step
from lcp_physics_amd.physics.batched_world import fused_step_backward
sc = scenes.make_stack_scenes(B=64, nbox=2, pts_per_interface=2, seed=21, dtype=torch.float32).to(device='cuda')
for path in ("big", "auto"):
    for (eps, mi) in ((1e-12, 10), (0.0, 10), (0.0, 14)):
        _lib.set_path(path)
        out = fused_step(sc, eps=eps, max_iter=mi); torch.cuda.synchronize()
        g = fused_step_backward(sc, out, torch.ones(64, sc.nb, 3, device='cuda')); torch.cuda.synchronize()
        _lib.set_path("auto")
        print(path, "eps", eps, "max_iter", mi, "mean iters %.2f" % float(out["iters"].float().mean()), "status!=0", int((out["status"] != 0).sum()),
              "scenes with NaN gradients", int(torch.isnan(g["v"]).reshape(64, -1).any(dim=1).sum()), "min s %.1e" % float(out["s"][out["s"] > 0].min()))

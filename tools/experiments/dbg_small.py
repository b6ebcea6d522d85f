import sys, torch
sys.path.insert(0, '/root/repo')
from lcp_physics_amd import _lib, scenes
from lcp_physics_amd.physics import fused_step
from lcp_physics_amd.physics.batched_world import fused_step_backward
for (nbox, pts, B) in ((2, 2, 8), (4, 2, 8), (4, 4, 3), (2, 4, 1)):
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=21, dtype=torch.float32).to(device='cuda')
    res = {}
    for path in ("auto", "big"):
        _lib.set_path(path)
        out = fused_step(sc); torch.cuda.synchronize()
        cot = torch.ones(B, sc.nb, 3, device='cuda')
        g = fused_step_backward(sc, out, cot); torch.cuda.synchronize()
        _lib.set_path("auto")
        res[path] = (out, g)
    a, b = res["auto"][0], res["big"][0]
    print((nbox, pts, B), "v_new diff", float((a["v_new"] - b["v_new"]).abs().max()), "nan in v", bool(torch.isnan(a["v_new"]).any()), "z diff", float((a["z"] - b["z"]).abs().max()),
          "s diff", float((a["s"] - b["s"]).abs().max()), "iters", a["iters"].tolist(), b["iters"].tolist(), "status", a["status"].tolist(), b["status"].tolist())
    ga, gb = res["auto"][1], res["big"][1]
    print("    grad v: nan", bool(torch.isnan(ga["v"]).any()), "diff", float((ga["v"] - gb["v"]).abs().max()))
sc = scenes.make_stack_scenes(B=8, nbox=2, pts_per_interface=2, seed=21, dtype=torch.float32).to(device='cuda')
for path in ("auto", "big"):
    _lib.set_path(path)
    out = fused_step(sc); torch.cuda.synchronize()
    g = fused_step_backward(sc, out, torch.ones(8, sc.nb, 3, device='cuda')); torch.cuda.synchronize()
    _lib.set_path("auto")
    m = 4 * sc.nc
    print(path, "min s per scene", out["s"].reshape(8, -1).min(dim=1)[0].tolist())
    print(path, "min z per scene", out["z"].reshape(8, -1).min(dim=1)[0].tolist())
    print(path, "nan grads per scene", torch.isnan(g["v"]).reshape(8, -1).any(dim=1).tolist())

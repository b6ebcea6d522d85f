"""Small batches: four scenes per wave (lcp_quad.hip) leave SIMDs idle below 4096 scenes; one wave per scene (lcp_primal.hip) ?"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from lcp_physics_amd import _lib, scenes
from lcp_physics_amd.physics import fused_step
for (nbox, pts) in ((2, 4), (4, 4)):
    for B in (256, 512, 1024, 2048, 4096, 8192):
        sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=5, dtype=torch.float32).to('cuda')
        res = {}
        for path in ("auto", "primal"):
            _lib.set_path(path)
            out = fused_step(sc); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(20): out = fused_step(sc, ws=out["ws"], out=out)
            torch.cuda.synchronize()
            res[path] = ((time.perf_counter() - t) / 20, out["v_new"].clone())
            _lib.set_path("auto")
        print("nbox %d pts %d B %5d: quad %.4f ms  wave-per-scene %.4f ms   max |dv| %.1e" % (nbox, pts, B, res["auto"][0] * 1e3, res["primal"][0] * 1e3,
              float((res["auto"][1] - res["primal"][1]).abs().max())))

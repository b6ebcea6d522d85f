"""Latency of the contact-list forward against the batch size: four scenes per wavefront (lcp_quad.hip, path "quad") and one scene
per wavefront (lcp_solo.hip, path "solo"), forward only, scenes resident in HBM.
    python tools/bench_batch_curve.py [nbox] > profiles/r03_batch_curve_<nbox>box.json"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lcp_physics_amd import scenes
from lcp_physics_amd.physics import fused_step

nbox = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = 200
rows = []
for B in (256, 512, 1024, 1536, 2048, 3072, 4096, 8192):
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=4, seed=1236, dtype=torch.float32).to(device="cuda")
    row = {"B": B}
    for path in ("quad", "solo", "auto"):
        out = None
        for _ in range(20):
            out = fused_step(sc, out=out, ws=None if out is None else out["ws"], path=path)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fused_step(sc, out=out, ws=out["ws"], path=path)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row[path + "_ms"] = ms
        row[path + "_steps_per_s"] = B / (ms * 1e-3)
    rows.append(row)
    print("B=%5d  quad %.4f ms (%.1f M/s)   solo %.4f ms (%.1f M/s)   auto %.4f ms" % (
        B, row["quad_ms"], row["quad_steps_per_s"] / 1e6, row["solo_ms"], row["solo_steps_per_s"] / 1e6, row["auto_ms"]), file=sys.stderr)
print(json.dumps({"what": "lcp_step_fused_f32 forward only, %d-box stacks (%d contacts), 4 pts/interface, fp64 arithmetic; events over %d launches"
                          % (nbox, 4 * nbox, reps), "rows": rows}, indent=1))

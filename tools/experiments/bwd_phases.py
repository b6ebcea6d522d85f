"""How the dense backward's time splits: everything, dp only (load + solve, no outer-product stores), dF only removed ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lcp_physics_amd import scenes
from lcp_physics_amd.lcp import lcp_backward
from lcp_physics_amd.physics import assemble_contacts, fused_step
from lcp_physics_amd.physics.batched_world import solution_of_step
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=1236, dtype=torch.float32).to("cuda")
lcp = assemble_contacts(sc)
out = fused_step(sc)
sol = solution_of_step(sc, out, lcp[2], lcp[4])
cot = torch.randn(B, 15, device="cuda")
def t(need, reps=200):
    g = lcp_backward(sol, cot, need=need)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lcp_backward(sol, cot, need=need, out=g)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
names = "Q p G h A b F".split()
for label, need in (("all seven", (True,) * 7), ("dp only", (False, True) + (False,) * 5), ("all but dF", (True,) * 6 + (False,)),
                    ("dF only", (False,) * 6 + (True,)), ("dG only", (False, False, True, False, False, False, False))):
    print("%-12s %.2f us" % (label, t(need)))

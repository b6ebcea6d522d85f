import sys, torch
sys.path.insert(0, '/root/repo')
from lcp_physics_amd import _lib, scenes
from lcp_physics_amd.lcp import lcp_backward
from lcp_physics_amd.physics.batched_world import assemble_contacts, fused_step, fused_step_backward, solution_of_step, solve_dynamics
from lcp_physics_amd.physics.contacts import ContactBuffers
from oracle import pdipm_oracle as O
from tests import parity
B=32
sc = scenes.make_stack_scenes(B=B, nbox=6, pts_per_interface=4, seed=546, dtype=torch.float32)
scg = sc.to(device='cuda')
cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(5), dtype=torch.float32).cuda()
res = {}
for path in ("auto", "big"):
    _lib.set_path(path)
    out = fused_step(scg)
    res[path] = {k: v.double().cpu() for k, v in fused_step_backward(scg, out, cot).items()}
    res[path + "_out"] = out
    _lib.set_path("auto")
gen = fused_step(scg, path="generic")
lcp = assemble_contacts(scg)
dense = lcp_backward(solution_of_step(scg, gen, lcp[2], lcp[4]), (-cot).reshape(B, -1))
torch.cuda.synchronize()
dense = {k: (None if t is None else t.double().cpu()) for k, t in zip("QpGhAbF", dense)}
ph = {k: v.double() if v.is_floating_point() else v for k, v in sc.phys_dict().items()}
ref = parity.physical_grads(ph, sc.dt, dense, O)
k = 1
torch.set_printoptions(precision=4, linewidth=200)
print("f grad scene 1: primal", res["auto"]["f"][k].flatten()[9:])
print("                big   ", res["big"]["f"][k].flatten()[9:])
print("                generic dense", ref["f"][k].flatten()[9:])
for path in ("auto", "big"):
    o = res[path + "_out"]
    print(path, "iters", int(o["iters"][k]), "status", int(o["status"][k]))
    z = o["z"][k].double().cpu(); s = o["s"][k].double().cpu()
    print("   min s/z", float((s / z).min()), "max", float((s/z).max()), " z range", float(z.min()), float(z.max()), " s range", float(s.min()), float(s.max()))
print("generic iters", int(gen["iters"][k]))
print("v_new diff primal-big", float((res["auto_out"]["v_new"][k] - res["big_out"]["v_new"][k]).abs().max()), "primal-generic", float((res["auto_out"]["v_new"][k] - gen["v_new"][k]).abs().max()))

d = (res["auto"]["f"] - res["big"]["f"]).abs().reshape(B,-1)
print("worst entries per scene (primal - big)", d.max(dim=1)[0][:6], "argmax scene1", int(d[1].argmax()))
d2 = (res["auto"]["f"] - ref["f"]).abs().reshape(B,-1); d3 = (res["big"]["f"] - ref["f"]).abs().reshape(B,-1)
print("primal - generic", float(d2[1].max()), " big - generic", float(d3[1].max()), " scale", float(ref["f"][1].abs().max()))

import sys, torch
sys.path.insert(0, '/root/repo')
from lcp_physics_amd import _lib, scenes
from lcp_physics_amd.physics import assemble_contacts, fused_step
from oracle import pdipm_oracle as O
B=128
sc = scenes.make_stack_scenes(B=B, nbox=2, pts_per_interface=4, seed=21, dtype=torch.float32)
scg = sc.to(device='cuda')
lcp = [None if t is None else t.double().cpu() for t in assemble_contacts(scg)]
ref = O.lcp_forward(*lcp)
res = {}
for path in ("auto", "big"):
    _lib.set_path(path)
    out = fused_step(scg); torch.cuda.synchronize()
    _lib.set_path("auto")
    res[path] = out
k, r = 59, 26
for path in res:
    o = res[path]
    print(path, "z", float(o["z"][k, r]), "s", float(o["s"][k, r]), "iters", int(o["iters"][k]), "status", int(o["status"][k]))
print("oracle z", float(ref.z[k, r]), "s", float(ref.s[k, r]), "iters", int(ref.iters[k]))
print("v_new diff auto-big", float((res["auto"]["v_new"][k] - res["big"]["v_new"][k]).abs().max()), "auto-oracle", float((res["auto"]["v_new"][k].double().cpu().flatten() + ref.x[k]).abs().max()))
print("max |z_auto - z_big| over batch", float((res["auto"]["z"] - res["big"]["z"]).abs().max()), " |s| ", float((res["auto"]["s"] - res["big"]["s"]).abs().max()), "iters differ", int((res["auto"]["iters"] != res["big"]["iters"]).sum()))

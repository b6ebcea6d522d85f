import sys, torch
sys.path.insert(0, '/root/repo')
from tests import test_hip_primal as T
from lcp_physics_amd import _lib
from lcp_physics_amd.physics.batched_world import fused_step_backward
B = 32
sc = T._scenes((6, 4), B)
count = torch.full((B,), sc.nc, dtype=torch.int32)
cot = torch.randn(B, sc.nb, 3, generator=torch.Generator().manual_seed(5), dtype=torch.float32).to("cuda")
for rep in range(3):
    grads = {}; outs = {}
    for path in ("auto", "big"):
        scg, out = T._solve(sc, count, path)
        _lib.set_path(path)
        grads[path] = {k: v.double().cpu() for k, v in fused_step_backward(scg, out, cot).items()}
        torch.cuda.synchronize()
        _lib.set_path("auto")
        outs[path] = out
    d = (grads["auto"]["f"] - grads["big"]["f"]).abs().reshape(B, -1).max(dim=1)[0]
    print("rep", rep, "max diff per scene", d.max().item(), "at", int(d.argmax()), " v_new diff", float((outs["auto"]["v_new"] - outs["big"]["v_new"]).abs().max()),
          "iters", outs["auto"]["iters"][int(d.argmax())].item(), outs["big"]["iters"][int(d.argmax())].item())
k = int(d.argmax())
torch.set_printoptions(precision=4, linewidth=220)
print(grads["auto"]["f"][k].flatten()); print(grads["big"]["f"][k].flatten())
z = outs["auto"]["z"][k].double().cpu(); s = outs["auto"]["s"][k].double().cpu()
print("s/z sorted low", torch.sort(s / z)[0][:10])

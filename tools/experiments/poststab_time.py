import sys, time, torch
sys.path.insert(0, '/root/repo')
from lcp_physics_amd import scenes
from lcp_physics_amd.physics.batched_world import post_stabilization, solve_dynamics
from lcp_physics_amd.physics.contacts import ContactBuffers
for (nbox, pts, B) in ((4, 4, 4096), (2, 4, 4096)):
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=5, dtype=torch.float32).to('cuda')
    cb = ContactBuffers(B, sc.nb, sc.nc, 'cuda')
    cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2
    count = torch.full((B,), sc.nc, dtype=torch.int32, device='cuda')
    out = post_stabilization(B, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.rest, cb, sc.Je); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): out = post_stabilization(B, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.rest, cb, sc.Je, ws=out["ws"], out=out)
    torch.cuda.synchronize()
    print("post_stabilization B %d nb %d nc %d: %.3f ms  (%.2f M scenes/s), mean iters %.1f" % (B, sc.nb, sc.nc, (time.perf_counter() - t) / 5 * 1e3, B / ((time.perf_counter() - t) / 5) / 1e6, float(out["iters"].float().mean())))
# against the generic kernel (forced path) on the same scenes
from lcp_physics_amd import _lib
sc = scenes.make_stack_scenes(B=256, nbox=4, pts_per_interface=4, seed=7, dtype=torch.float32).to('cuda')
sc.v += 0.3 * torch.randn_like(sc.v)
cb = ContactBuffers(256, sc.nb, sc.nc, 'cuda')
cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2
count = torch.randint(0, sc.nc + 1, (256,), dtype=torch.int32, device='cuda')
a = post_stabilization(256, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.rest, cb, sc.Je)
_lib.set_path("generic")
b = post_stabilization(256, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.rest, cb, sc.Je)
_lib.set_path("auto")
torch.cuda.synchronize()
d = (a["dp"] - b["dp"]).abs().reshape(256, -1).max(dim=1)[0]
print("body space vs generic: max |dp - dp_generic| %.2e (|dp| max %.2e), iters differ in %d scenes, status %s %s" % (float(d.max()), float(b["dp"].abs().max()), int((a["iters"] != b["iters"]).sum()), a["status"].unique().tolist(), b["status"].unique().tolist()))

"""Diagnostic: the eager differentiable roll-out at 4096 scenes - seconds per iteration and what the caching allocator does meanwhile."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_hip_contacts import _rollout_world
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = np.load(os.path.join(ROOT, "tests", "golden", "rollout_grad.npz"))
world, force0 = _rollout_world(d, rep)
p0, v0 = world.p.clone(), world.v.clone()
i, j = [int(k) for k in d["loss_bodies"]]
nsteps = int(d["nsteps"])
def loss_of():
    world.restart(p0, v0)
    for _ in range(nsteps):
        world.step(differentiable=True)
    pos = world.p[:, :, 1:]
    return (pos[:, i] - pos[:, j]).norm(dim=1)
import gc
for it in range(8):
    if it >= 4:
        n = gc.collect()                          # (second half: with a collection before every iteration)
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    force0.grad = None
    l = loss_of(); torch.cuda.synchronize(); t1 = time.perf_counter()
    l.sum().backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print("gc " if it >= 4 else "   ", end="")
    print("iter %d: forward %.4f s backward %.4f s | device mallocs %d frees %d retries %d | reserved %.2f GB allocated peak %.2f GB" % (
        it, t1 - t0, t2 - t1, st["num_device_alloc"] - st0["num_device_alloc"], st["num_device_free"] - st0["num_device_free"],
        st["num_alloc_retries"] - st0["num_alloc_retries"], st["reserved_bytes.all.current"] / 2**30, st["allocated_bytes.all.peak"] / 2**30))
import collections
big = [o for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.is_cuda and o.numel() * o.element_size() > 50e6]
print("live CUDA tensors > 50 MB:", len(big), "current allocated %.2f GB" % (torch.cuda.memory_allocated() / 2**30))
cnt = collections.Counter()
for t in big[:400]:
    for r in gc.get_referrers(t):
        if r is big: continue
        cnt[type(r).__name__ + (":" + ",".join(sorted(k for k in r.keys() if isinstance(k, str))[:8]) if isinstance(r, dict) else "")] += 1
for k, v in cnt.most_common(12): print("  referrer", v, k[:200])
# one level up: who refers to the dicts that hold a workspace
holders = [r for t in big[:40] for r in gc.get_referrers(t) if isinstance(r, dict) and "ws" in r]
cnt2 = collections.Counter()
for h in holders[:40]:
    for r in gc.get_referrers(h):
        if r is holders: continue
        cnt2[type(r).__name__ + (":" + ",".join(sorted(str(k) for k in r.keys())[:8]) if isinstance(r, dict) else "")] += 1
for k, v in cnt2.most_common(12): print("  holder of a ws-dict", v, k[:200])
from lcp_physics_amd import _lib
print("workspace bytes per scene:", _lib.workspace_bytes(1, 3 * world.nb, 4 * world.maxc, world.e, 1), "nb", world.nb, "maxc", world.maxc, "e", world.e)

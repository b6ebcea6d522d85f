"""Round 6, chain root cause: run the 8-2-16 chain case (max_iter = 1: the initialisation pass) on a library whose <56>-column kernel
carries assembly-level register dumps (tools/asm_dump_instrument.py) and save the dumps of the first scenes."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_hip_primal import _with_joint_rows                    # noqa: E402
from lcp_physics_amd import scenes                                    # noqa: E402
from lcp_physics_amd.physics.batched_world import solve_dynamics      # noqa: E402
from lcp_physics_amd.physics.contacts import ContactBuffers           # noqa: E402

tag = sys.argv[1]
B, nbox, pts, e = 32, 8, 2, 16
NS, NSLOT, NROW = 4, 10, 640
sc = _with_joint_rows(scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=pts, seed=900 + nbox + e, dtype=torch.float32), e)
scg = sc.to(device="cuda")
cb = ContactBuffers(sc.B, sc.nb, sc.nc, "cuda")
cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = scg.c_n, scg.c_p1, scg.c_p2, scg.c_i1, scg.c_i2
count = torch.full((B,), sc.nc, dtype=torch.int32, device="cuda")
ws = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
out = solve_dynamics(sc.B, sc.nb, sc.nc, e, count, scg.Mdiag, scg.v, scg.f, scg.rest, scg.fric, cb, scg.Je, sc.dt, max_iter=1, ws=ws)
torch.cuda.synchronize()
ptr = ws.data_ptr()
base = ((ptr + 0x1fffffff) & ~0x0fffffff) - ptr
dump = np.zeros((NS, NSLOT, NROW, 64), dtype=np.int32)
for s in range(NS):
    for k in range(NSLOT):
        o = base + (s << 22) + (k << 18)
        dump[s, k] = ws[o: o + NROW * 256].view(torch.int32).reshape(NROW, 64).cpu().numpy()
it = ws.view(torch.float64)[: B * 792].reshape(B, 792).cpu().numpy()
np.savez_compressed("gpurun_out/r06_chain_regdump_%s.npz" % tag, dump=dump, ws1=it, iters=out["iters"].cpu().numpy())
print(tag, "tags of scene 0:", dump[0, :, 0, 0].tolist(), "x[12..16] of scene 0:", it[0, 64 + 12: 64 + 16].tolist())

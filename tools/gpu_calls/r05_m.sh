#!/bin/bash
# round 5, call m: post-stabilisation on the four-scenes-per-wave mapping (lcp_fwd_quad<..., POST>): tests that touch it + timing against the one-wave-per-scene kernel
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_primal.py tests/test_hip_contacts.py tests/test_hip_parity.py -q -m gpu -k "stab or poststab or post_stab or chain or engine" 2>&1 | grep -E "passed|failed|^FAILED|Error" | cut -c1-300
timeout 300 python tools/experiments/poststab_time.py 2>/dev/null | tee $O/r05_m_poststab_time.txt
python - <<'PY' 2>/dev/null | tee -a gpurun_out/r05_m_poststab_time.txt
import sys, time, torch
sys.path.insert(0, '.')
from lcp_physics_amd import scenes, _lib
from lcp_physics_amd.physics.batched_world import post_stabilization
from lcp_physics_amd.physics.contacts import ContactBuffers
B = 4096
sc = scenes.make_stack_scenes(B=B, nbox=4, pts_per_interface=4, seed=5, dtype=torch.float32).to('cuda')
cb = ContactBuffers(B, sc.nb, sc.nc, 'cuda')
cb.c_n, cb.c_p1, cb.c_p2, cb.c_i1, cb.c_i2 = sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2
count = torch.full((B,), sc.nc, dtype=torch.int32, device='cuda')
res = {}
for path in ("auto", "primal"):
    _lib.set_path(path)
    out = post_stabilization(B, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.rest, cb, sc.Je); torch.cuda.synchronize()
    for _ in range(10): out = post_stabilization(B, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.rest, cb, sc.Je, ws=out["ws"], out=out)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50): out = post_stabilization(B, sc.nb, sc.nc, 3, count, sc.Mdiag, sc.v, sc.rest, cb, sc.Je, ws=out["ws"], out=out)
    ev[1].record(); torch.cuda.synchronize()
    res[path] = (ev[0].elapsed_time(ev[1]) / 50, out["dp"].clone(), out["iters"].clone())
    print("post_stabilization 4096 x 16, path %-7s %.4f ms per launch, mean iters %.2f" % (path, res[path][0], float(out["iters"].float().mean())))
_lib.set_path("auto")
d = (res["auto"][1] - res["primal"][1]).abs().max()
print("four scenes per wave vs one wave per scene: max |dp - dp'| %.3e (|dp| max %.3e), iteration counts differ on %d scenes" % (float(d), float(res["primal"][1].abs().max()), int((res["auto"][2] != res["primal"][2]).sum())))
PY

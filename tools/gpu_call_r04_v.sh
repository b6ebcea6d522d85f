#!/bin/bash
bench() { timeout 300 python bench.py --no-cpu-baseline --no-companions $2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$1 %s: %.2f M  ms/step %.4f  fwd %.4f ms  bwd %.4f ms' % ('$2', j['value']/1e6, j['ms_per_step'], r['fwd_ms'], r['bwd_ms']))"; }
for i in 1 2 3; do bench "HEAD          "; LCP_HIP_LIB=$PWD/tools/liblcp_exp_unr.so bench "passes unrolled"; done
bench "HEAD           " "--mode dense"; LCP_HIP_LIB=$PWD/tools/liblcp_exp_unr.so bench "passes unrolled" "--mode dense"
bench "HEAD           " "--batch 32768"; LCP_HIP_LIB=$PWD/tools/liblcp_exp_unr.so bench "passes unrolled" "--batch 32768"
python - <<'PY'
import os, torch
from lcp_physics_amd import scenes
from lcp_physics_amd.physics import fused_step
sc = scenes.make_stack_scenes(B=4096, nbox=4, pts_per_interface=4, seed=1236, dtype=torch.float32).to(device="cuda")
a = fused_step(sc)
torch.save({k: a[k].cpu() for k in ("v_new", "z", "s", "iters")}, "/tmp/head.pt")
PY
LCP_HIP_LIB=$PWD/tools/liblcp_exp_unr.so python - <<'PY'
import torch
from lcp_physics_amd import scenes
from lcp_physics_amd.physics import fused_step
sc = scenes.make_stack_scenes(B=4096, nbox=4, pts_per_interface=4, seed=1236, dtype=torch.float32).to(device="cuda")
a = fused_step(sc); h = torch.load("/tmp/head.pt")
print("bitwise equal to HEAD:", {k: bool(torch.equal(a[k].cpu(), h[k])) for k in h})
PY

#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python tools/experiments/rollout_alloc_diag.py 512 2>&1 | grep -v "amdgpu\|Warning\|big = " | head -12 > $O/r04_rollout_leak_after.txt; cat $O/r04_rollout_leak_after.txt | cut -c1-200
timeout 900 python -m pytest tests/test_hip_contacts.py tests/test_hip_parity.py -q -m gpu -x -k "release or rollout or graph or engine_plugin or lcpfunction or differentiable" > $O/r04_leak_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r04_leak_tests.log
{ timeout 300 python tools/experiments/grad_demo_rollout.py --rep 128 --eager --count; timeout 300 python tools/experiments/grad_demo_rollout.py --rep 128; timeout 300 python tools/experiments/grad_demo_rollout.py --rep 512 --eager; timeout 300 python tools/experiments/grad_demo_rollout.py --rep 512; } 2>/dev/null | grep "^{" > $O/r04_grad_demo_rollout.json; cut -c1-330 $O/r04_grad_demo_rollout.json
{ timeout 300 python tools/experiments/mass_inference.py --batch 4096 --count; timeout 300 python tools/experiments/mass_inference.py --batch 4096 --graph; } 2>$O/r04_mass_inference.err | grep "^{" > $O/r04_mass_inference.json; cut -c1-200 $O/r04_mass_inference.json; tail -3 $O/r04_mass_inference.err

#!/bin/bash
# round 6: lcp_solo.o built with the iterative-ilp scheduler strategy: parity of everything that runs on it, then its bench lines and phase profile
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_solo.py tests/test_hip_contacts.py tests/test_hip_reentrancy.py tests/test_hip_step_backward.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_hip_headline_parity.py -m gpu -x -q -k "configs1" 2>&1 | tail -3
timeout 300 python bench.py --config 1 > gpurun_out/r06_bench_config1_parity.json 2>/dev/null; tail -1 gpurun_out/r06_bench_config1_parity.json | cut -c1-200
timeout 300 python bench.py --config 1 --no-cpu-baseline > gpurun_out/r06_bench_config2_fwd_only.json 2>/dev/null; tail -1 gpurun_out/r06_bench_config2_fwd_only.json | cut -c1-200
LCP_HIP_LIB=$PWD/tools/liblcp_soloprof.so timeout 200 python tools/gpu_phase_profile_solo.py > gpurun_out/r06_solo_phase_profile.txt 2>&1; cat gpurun_out/r06_solo_phase_profile.txt | tail -12
timeout 300 python tools/bench_batch_curve.py 2 > gpurun_out/r06_batch_curve_2box.json 2>/dev/null; tail -c 600 gpurun_out/r06_batch_curve_2box.json

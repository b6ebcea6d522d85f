#!/bin/bash
# round 6, final: the GPU suite at HEAD (lcp_solo.o with iterative-ilp), smoke(), the solo phase profile of the product's scheduling
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --durations=25 2>&1 | tail -45 > gpurun_out/r06_gputests.txt
tail -3 gpurun_out/r06_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r06_smoke.txt
LCP_HIP_LIB=$PWD/tools/liblcp_soloprof.so timeout 200 python tools/gpu_phase_profile_solo.py > gpurun_out/r06_solo_phase_profile.txt 2>&1; tail -11 gpurun_out/r06_solo_phase_profile.txt

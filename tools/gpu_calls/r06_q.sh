#!/bin/bash
# round 6, call q: in-kernel phase profiles of the solo kernel (configs[1]) and the quad kernel (headline)
O=gpurun_out; mkdir -p $O
LCP_HIP_LIB=$PWD/tools/liblcp_soloprof.so python tools/gpu_phase_profile_solo.py 1024 2 2>&1 | grep -v amdgpu.ids | tee $O/r06_solo_phase_profile.txt
LCP_HIP_LIB=$PWD/tools/liblcp_quadprof.so python tools/gpu_phase_profile_quad.py 4096 4 2>&1 | grep -v amdgpu.ids | tee $O/r06_quad_phase_profile.txt

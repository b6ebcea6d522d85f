#!/bin/bash
# round 6, call o: how the dense backward's time splits (which outputs cost what)
LCP_HIP_LIB=$PWD/lcp_physics_amd/csrc/variants/nosplit.so python tools/experiments/bwd_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_bwd_phases.txt

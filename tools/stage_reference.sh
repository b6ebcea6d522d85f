#!/bin/bash
# MEASUREMENT ONLY.  The build container has the reference and no GPU, the GPU box a GPU and no reference.
# For ONE gpurun call a read-only copy of the reference's python package is staged in untracked scratch (oracle/_ref/stage:
# git-ignored, so it can never be committed; it travels to the GPU box with the snapshot like a built .so), the measurement runs
# there, and the copy is removed again whatever happens.  Nothing of the product imports it (LCP_REFERENCE_ROOT is read by
# oracle/ref_shim.py only).  Outputs: gpurun_out/<TAG>_reference_world_plugin.json, gpurun_out/<TAG>_reference_cpu_timing_*.jsonl, gpurun_out/<TAG>_bench_fused_with_reference.json.
set -e
cd "$(dirname "$0")/.."
STAGE=oracle/_ref/stage
rm -rf $STAGE; mkdir -p $STAGE
trap 'rm -rf oracle/_ref/stage' EXIT
cp -r /root/reference/lcp_physics $STAGE/
find $STAGE -name "__pycache__" -prune -exec rm -rf {} +
/usr/local/graft/bin/gpurun --timeout ${TIMEOUT:-1500} -- "${CMD:-bash tools/gpu_reference_measure.sh}"

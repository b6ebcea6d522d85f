#!/bin/bash
# final sanity at HEAD: smoke(), the default bench line, the driver's 20-step form, 2 ranks through torchrun on one device
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-devices --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['n_gpus'], d['value'], d['config'].get('control_plane'))"

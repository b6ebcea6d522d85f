#!/bin/bash
# round 5, call f: the whole GPU suite (chain unit back on IEEE quotients), the two-rank protocol with RCCL tried on ONE device (it must fall back to gloo and say so)
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r05_f_gputests.log 2>&1; echo "suite rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" $O/r05_f_gputests.log | tail -8 | cut -c1-400
timeout 300 python bench.py --gpus 2 --share-devices --steps 20 --warmup 5 --no-cpu-baseline > $O/r05_f_bench_2ranks_shared.json 2> $O/r05_f_bench_2ranks_shared.err; echo "2 ranks rc=$?"; tail -1 $O/r05_f_bench_2ranks_shared.json | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['value'], j['n_gpus'], j.get('devices_used'), j['config'].get('control_plane'))"
tail -3 $O/r05_f_bench_2ranks_shared.err | cut -c1-300

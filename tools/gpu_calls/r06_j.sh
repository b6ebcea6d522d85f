#!/bin/bash
# round 6, call j: the eight ranks of the driver's scaling run, rehearsed on ONE device (RCCL refuses -> gloo; eight processes contending for one host and one GPU)
O=gpurun_out; mkdir -p $O
timeout 900 python bench.py --gpus 8 --share-devices --no-cpu-baseline > $O/r06_bench_8ranks_one_device.json 2> $O/r06_bench_8ranks_one_device.err; echo rc $?; tail -1 $O/r06_bench_8ranks_one_device.json | cut -c1-400; tail -3 $O/r06_bench_8ranks_one_device.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --share-devices --steps 20 --warmup 5 > $O/r06_bench_2ranks_torchrun.json 2> $O/r06_bench_2ranks_torchrun.err; echo rc $?; tail -1 $O/r06_bench_2ranks_torchrun.json | cut -c1-300
timeout 300 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_form_j.json 2>/dev/null; tail -1 $O/r06_bench_driver_form_j.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['host_us_per_step'], d['host_bound_ceiling']['value'], d['roofline'].get('counters_stale'))"

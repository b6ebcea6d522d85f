#!/bin/bash
# round 6, call m: is the body-space backward solve bound by instruction fetch ?  (the solve repeated in a loop; one copy of the KKT solve)
O=gpurun_out; mkdir -p $O
for pass in 1 2; do STEPS=200 bash tools/ab_bench.sh; done 2>&1 | tee $O/r06_ab_bwd_fetch.txt

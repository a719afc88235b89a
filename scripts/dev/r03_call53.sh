#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
for rep in 1 2 3; do
for v in lw8 lw4; do
    r=$(LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so LIPVQ_SCREEN_MONITOR=0 timeout -k 10 200 python bench.py --workload cfg2 --metric-only --no-cpu-baseline --sustained 0 --steps 50 --warmup 20 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$v cfg2 $r" | tee -a $O/lane_w4_ab.txt
done
done

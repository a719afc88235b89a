#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
for rep in 1 2 3; do
for v in base ring42 ring43; do
    r=$(LIPVQ_SCREEN_MONITOR=0 LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so timeout -k 10 200 python bench.py --workload cfg2 --metric-only --no-cpu-baseline --sustained 0 --steps 50 --warmup 20 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$v cfg2 $r" | tee -a $O/ring_ab.txt
done
done
LIPVQ_HIP_LIBRARY=build_ab/ring42/_lipvq_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -x -q -k "equals_oracle_and_unfused or shapes" 2>&1 | tail -2

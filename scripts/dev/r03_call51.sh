#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
for rep in 1 2 3; do
for v in lw_old lw_new; do
  for wl in cfg3 icrt; do
    r=$(LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so LIPVQ_SCREEN_MONITOR=0 timeout -k 10 200 python bench.py --workload $wl --metric-only --no-cpu-baseline --sustained 0 --steps 40 --warmup 15 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$v $wl $r" | tee -a $O/lane_w_ab.txt
  done
done
done
LIPVQ_HIP_LIBRARY=build_ab/lw_new/_lipvq_hip.so timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_big_parity.py -x -q 2>&1 | tail -2

#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
for rep in 1 2; do
for v in base nostage nobar noldsb; do
  for wl in cfg2 cfg3; do
    r=$(LIPVQ_SCREEN_MONITOR=0 LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so timeout -k 10 200 python bench.py --workload $wl --metric-only --no-cpu-baseline --sustained 0 --steps 50 --warmup 20 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$v $wl $r" | tee -a $O/ablate_after.txt
  done
done
done

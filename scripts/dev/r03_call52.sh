#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
for rep in 1 2; do
for v in base minreg; do
  for wl in cfg2 cfg3 icrt; do
    r=$(LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so LIPVQ_SCREEN_MONITOR=0 timeout -k 10 200 python bench.py --workload $wl --metric-only --no-cpu-baseline --sustained 0 --steps 40 --warmup 15 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$v $wl $r" | tee -a $O/minreg_ab.txt
  done
done
done

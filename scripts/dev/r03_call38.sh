#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
for rep in 1 2; do
for v in base after after4; do
  for wl in cfg2 cfg3 icrt; do
    for m in fine coarse; do
    r=$(LIPVQ_SCREEN_MODE=$m LIPVQ_SCREEN_MONITOR=0 LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so timeout -k 10 200 python bench.py --workload $wl --metric-only --no-cpu-baseline --sustained 0 --steps 50 --warmup 20 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$v $wl $m $r" | tee -a $O/frag_after_ab.txt
    done
  done
done
done
LIPVQ_HIP_LIBRARY=build_ab/after/_lipvq_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_screen.py tests/test_gpu_big_parity.py -x -q 2>&1 | tail -2

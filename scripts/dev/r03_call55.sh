#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
for rep in 1 2 3; do
for v in head cur; do
  lib=build_ab/head/_lipvq_hip.so; [ $v = cur ] && lib=lipvq-vae_amd/_lipvq_hip.so
  for wl in cfg3 icrt; do
    r=$(LIPVQ_HIP_LIBRARY=$lib LIPVQ_SCREEN_MONITOR=0 timeout -k 10 200 python bench.py --workload $wl --metric-only --no-cpu-baseline --sustained 0 --steps 40 --warmup 15 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$v $wl $r" | tee -a $O/load_x_ab.txt
  done
done
done
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_big_parity.py -x -q 2>&1 | tail -2

#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
for rep in 1 2; do
for v in nopin pin; do
  lib=build_ab/nopin/_lipvq_hip.so; [ $v = pin ] && lib=lipvq-vae_amd/_lipvq_hip.so
  echo "== $v" | tee -a $O/wgrad_pin_ab.txt
  LIPVQ_HIP_LIBRARY=$lib timeout -k 10 300 python scripts/measure_wgrad.py 2>&1 | grep "wgrad N=524288" | tee -a $O/wgrad_pin_ab.txt
  LIPVQ_HIP_LIBRARY=$lib timeout -k 10 200 python scripts/dev/measure_train_big.py llfq 2>&1 | grep "train step" | tee -a $O/wgrad_pin_ab.txt
done
done
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad" 2>&1 | tail -2

#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
for rep in 1 2 3; do
for v in old new; do
  lib=build_ab/mlp_old/_lipvq_hip.so; [ $v = new ] && lib=lipvq-vae_amd/_lipvq_hip.so
  LIPVQ_HIP_LIBRARY=$lib timeout -k 10 200 python scripts/dev/measure_train_big.py llfq 2>&1 | grep "train step" | sed "s/^/$v /" | tee -a $O/mlpl_lane_ab.txt
done
done
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kernels.py tests/test_gpu_module.py -x -q 2>&1 | tail -2

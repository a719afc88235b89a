#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_module.py tests/test_gpu_icl.py -x -q > $O/test_vq2.txt 2>&1 || { tail -30 $O/test_vq2.txt; exit 1; }
tail -2 $O/test_vq2.txt
for k in 128 1024; do VQ_K=$k timeout -k 10 200 python scripts/dev/measure_train_big.py vq 2>&1 | grep "train step" | tee -a $O/train_vq2.txt; done
timeout -k 10 200 python scripts/dev/measure_vqvae.py 2>&1 | grep -v amdgpu | tee -a $O/train_vq2.txt
timeout -k 10 300 python scripts/dev/soak_train.py 120 2>&1 | grep -v amdgpu | tee -a $O/train_vq2.txt

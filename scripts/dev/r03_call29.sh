#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_module.py -x -q -k "vq" > $O/test_vq.txt 2>&1 || { tail -30 $O/test_vq.txt; exit 1; }
tail -3 $O/test_vq.txt
for cfg in "128 x" "1024 x" "1024 icrt" "8192 cfg3"; do
set -- $cfg
VQ_K=$1 LIPVQ_VQ_TRAIN_UNFUSED=1 timeout -k 10 200 python scripts/dev/measure_train_big.py vq $2 2>&1 | grep "train step" | tee -a $O/train_vq.txt
VQ_K=$1 timeout -k 10 200 python scripts/dev/measure_train_big.py vq $2 2>&1 | grep "train step" | tee -a $O/train_vq.txt
done

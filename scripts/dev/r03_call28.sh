#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -x -q -k "folded or formed_in_kernel" > $O/test_folded.txt 2>&1 || { tail -30 $O/test_folded.txt; exit 1; }
tail -3 $O/test_folded.txt
for wl in cfg3 icrt; do
LIPVQ_NO_FOLD=1 LIPVQ_TORCH_ADAMW=1 timeout -k 10 200 python scripts/dev/measure_train_big.py llfq $wl 2>&1 | grep "train step" | tee -a $O/train_big4.txt
timeout -k 10 200 python scripts/dev/measure_train_big.py llfq $wl 2>&1 | grep "train step" | tee -a $O/train_big4.txt
done

#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -x -q -k "folded or training_forward or no_grad_forward or formed_in_kernel" > $O/test_folded.txt 2>&1 || { tail -30 $O/test_folded.txt; exit 1; }
tail -3 $O/test_folded.txt
for k in llfq vq; do
LIPVQ_NO_FOLD=1 timeout -k 10 200 python scripts/dev/measure_train_big.py $k 2>&1 | grep "train step" | tee -a $O/train_big.txt
timeout -k 10 200 python scripts/dev/measure_train_big.py $k 2>&1 | grep "train step" | tee -a $O/train_big.txt
done
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/train -- python3 $OLDPWD/scripts/profile_train_step_big.py > /dev/null 2>&1 )
cp $(ls $O/train/*/*kernel_stats.csv | head -1) $O/kernel_stats_train_step_cfg2.csv
rm -rf $O/train
head -12 $O/kernel_stats_train_step_cfg2.csv | cut -c1-150

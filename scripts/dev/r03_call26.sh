#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_module.py tests/test_gpu_kernels.py tests/test_gpu_icl.py -x -q > $O/test_sel.txt 2>&1 || { tail -30 $O/test_sel.txt; exit 1; }
tail -3 $O/test_sel.txt
for k in llfq vq; do
LIPVQ_TORCH_ADAMW=1 timeout -k 10 200 python scripts/dev/measure_train_big.py $k 2>&1 | grep "train step" | sed 's/$/ torch AdamW/' | tee -a $O/train_big2.txt
timeout -k 10 200 python scripts/dev/measure_train_big.py $k 2>&1 | grep "train step" | tee -a $O/train_big2.txt
done
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/train -- python3 $OLDPWD/scripts/profile_train_step_big.py > /dev/null 2>&1 )
cp $(ls $O/train/*/*kernel_stats.csv | head -1) $O/kernel_stats_train_step_cfg2.csv
rm -rf $O/train
head -8 $O/kernel_stats_train_step_cfg2.csv | cut -c1-150
timeout -k 10 300 python scripts/dev/measure_mlp3_fwd.py 2>&1 | tail -8

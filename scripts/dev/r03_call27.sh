#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/test_gpu_all.txt 2>&1 || { tail -30 $O/test_gpu_all.txt; exit 1; }
tail -3 $O/test_gpu_all.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json | cut -c1-1500
for k in llfq vq; do timeout -k 10 200 python scripts/dev/measure_train_big.py $k 2>&1 | grep "train step" | tee -a $O/train_big3.txt; done
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/train -- python3 $OLDPWD/scripts/profile_train_step_big.py > /dev/null 2>&1 )
cp $(ls $O/train/*/*kernel_stats.csv | head -1) $O/kernel_stats_train_step_cfg2.csv
rm -rf $O/train

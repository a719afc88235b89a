#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/graph -- python3 $OLDPWD/scripts/dev/prof_graph_step.py > /dev/null 2>&1 )
cp $(ls $O/graph/*/*kernel_stats.csv | head -1) $O/kernel_stats_graphed_step_icrt.csv
rm -rf $O/graph
timeout -k 10 300 python tests/bench_train_step.py 2>&1 | grep -v amdgpu | tee $O/bench_train_step.txt

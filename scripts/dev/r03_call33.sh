#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
for n in 80 500; do
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/graph$n -- python3 $OLDPWD/scripts/dev/prof_graph_step.py $n 12 208 1024 > /dev/null 2>&1 )
cp $(ls $O/graph$n/*/*kernel_stats.csv | head -1) $O/kernel_stats_graphed_step_$n.csv
rm -rf $O/graph$n
done

#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03s; mkdir -p $O
export LIPVQ_SCREEN_MONITOR=0
for m in fine coarse; do
echo "== cfg2 $m" >> $O/stamps.txt
LIPVQ_SCREEN_MODE=$m LIPVQ_HIP_LIBRARY=build_ab/st_coarse/_lipvq_hip.so timeout -k 10 200 python scripts/stamps.py cfg2 2>&1 | grep -v amdgpu >> $O/stamps.txt
done
cat $O/stamps.txt
export TMPDIR=/tmp
LIPVQ_SCREEN_MODE=coarse rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg2 -- python3 bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline --sustained 0 --metric-only --traffic off > /dev/null 2> $O/prof_cfg2.err
cp $(ls $O/trace_cfg2/*/*kernel_stats.csv | head -1) $O/kernel_stats_cfg2_coarse.csv; head -5 $O/kernel_stats_cfg2_coarse.csv | cut -c1-60,180-250
rm -rf $O/trace_cfg2

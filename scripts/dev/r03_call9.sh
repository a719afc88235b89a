#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03i; mkdir -p $O
export LIPVQ_SCREEN_MONITOR=0
for wl in cfg2 cfg3 icrt; do for m in fine coarse; do
  echo "== $wl $m" >> $O/cand_stats.txt
  LIPVQ_SCREEN_MODE=$m timeout -k 10 200 python scripts/dev/cand_stats.py $wl 2>&1 | grep -v amdgpu >> $O/cand_stats.txt
done; done
cat $O/cand_stats.txt

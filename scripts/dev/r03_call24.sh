#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03x; mkdir -p $O
export LIPVQ_SCREEN_MONITOR=0
for m in fine coarse; do
echo "== icrt $m" >> $O/stamps_icrt.txt
LIPVQ_SCREEN_MODE=$m LIPVQ_HIP_LIBRARY=build_ab/st_final/_lipvq_hip.so timeout -k 10 200 python scripts/stamps.py icrt 2>&1 | grep -v amdgpu >> $O/stamps_icrt.txt
done
cat $O/stamps_icrt.txt

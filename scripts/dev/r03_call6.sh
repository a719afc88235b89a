#!/bin/bash
# GPU box, round 3 sixth call: the round's evidence set on the final build + the clock check behind the cfg3 "no LDS reads" ablation
set -e -o pipefail
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_f; mkdir -p $O
for v in st8_base st8_noldsb; do
  echo "== $v" >> $O/stamps_cfg3_clock.txt
  LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so timeout -k 10 200 python scripts/stamps.py cfg3 2>&1 | grep -v amdgpu | head -3 >> $O/stamps_cfg3_clock.txt
done
cat $O/stamps_cfg3_clock.txt
timeout -k 10 1000 bash scripts/dev/evidence.sh r03_f

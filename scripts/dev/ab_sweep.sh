#!/bin/bash
# GPU box: same-box A/B of library variants (build_ab/<name>/_lipvq_hip.so from scripts/dev/ab_one.sh; "main" = the in-tree build):
#   bash scripts/dev/ab_sweep.sh <outdir under gpurun_out> <workload> "<SWEEP_G list>" <passes> name1 name2 ...
set -e
cd "$(dirname "$0")/../.."
OUT=gpurun_out/$1; WL=$2; export SWEEP_G=$3; PASSES=$4; shift 4
mkdir -p $OUT
for p in $(seq 1 $PASSES); do
  for v in "$@"; do
    if [ "$v" = "main" ]; then unset LIPVQ_HIP_LIBRARY; else export LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so; fi
    echo "== $v pass $p" >> $OUT/ab_$WL.txt
    python scripts/dev/shard_sweep.py $WL 2>&1 | grep -E "^ +[0-9]" >> $OUT/ab_$WL.txt
  done
done
cat $OUT/ab_$WL.txt

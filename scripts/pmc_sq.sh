#!/bin/bash
# GPU box: SQ counters of the bench's kernels in two --pmc passes (8 SQ slots each), summarised per kernel.
# Usage: bash scripts/pmc_sq.sh <outdir under gpurun_out> [bench args...]
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --sustained 0 --metric-only $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d $OUT/pmc_a -- python3 bench.py $ARGS > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA \
  --output-format csv -d $OUT/pmc_b -- python3 bench.py $ARGS > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc_c -- python3 bench.py $ARGS > $OUT/pmc_c.log 2>&1 || echo "pass c failed (a counter name may not exist)"
python3 scripts/pmc_summary.py $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c > $OUT/sq_counters.txt
cat $OUT/sq_counters.txt

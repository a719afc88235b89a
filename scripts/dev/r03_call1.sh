#!/bin/bash
# GPU box, round 3 first call: new parity tests, shard-size sweep, cfg3 evidence, fast-mode bisect
set -e -o pipefail
cd "$(dirname "$0")/../.."
O=gpurun_out/r03a; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_screen.py tests/test_gpu_module.py -x -q -m gpu > $O/pytest_parity.txt 2>&1 || { tail -30 $O/pytest_parity.txt; exit 1; }
tail -3 $O/pytest_parity.txt
timeout -k 10 300 python scripts/dev/shard_sweep.py cfg2 > $O/shard_sweep_cfg2.txt 2>&1; cat $O/shard_sweep_cfg2.txt
timeout -k 10 300 python scripts/dev/shard_sweep.py cfg3 > $O/shard_sweep_cfg3.txt 2>&1; cat $O/shard_sweep_cfg3.txt
for t in a52b26b d17cef6 8ff6163 818d608 03e490e f5364ac 5c066b2 fd9a87e 4e5f492; do
  timeout -k 10 200 python scripts/dev/bisect_measure.py build_ab/trees/$t >> $O/fast_bisect.txt 2>&1 || echo "$t failed" >> $O/fast_bisect.txt
done
timeout -k 10 200 python scripts/dev/bisect_measure.py . >> $O/fast_bisect.txt 2>&1
cat $O/fast_bisect.txt
LIPVQ_HIP_LIBRARY=build_ab/stamps/_lipvq_hip.so timeout -k 10 200 python scripts/stamps.py cfg3 > $O/stamps_cfg3.txt 2>&1; cat $O/stamps_cfg3.txt
LIPVQ_HIP_LIBRARY=build_ab/stamps/_lipvq_hip.so timeout -k 10 200 python scripts/stamps.py cfg2 > $O/stamps_cfg2.txt 2>&1; cat $O/stamps_cfg2.txt
timeout -k 10 400 bash scripts/pmc_sq.sh r03a/sq_cfg3 --workload cfg3 > /dev/null 2>&1; cat $O/sq_cfg3/sq_counters.txt
timeout -k 10 300 bash scripts/pmc_hbm.sh r03a/hbm cfg3 > $O/hbm_cfg3.txt 2>&1; cat $O/hbm_cfg3.txt
rm -rf $O/sq_cfg3/pmc_* $O/hbm/pmc_*

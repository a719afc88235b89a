#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03k; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1 || { grep -E "^E|FAILED" $O/pytest_all.txt | head -20; }
tail -2 $O/pytest_all.txt
timeout -k 10 600 python scripts/dev/coarse_sweep.py 2>&1 | grep -v amdgpu > $O/coarse_sweep.txt; cat $O/coarse_sweep.txt
timeout -k 10 400 python bench.py --workload cfg3 > $O/bench_cfg3.json 2>$O/bench_cfg3.err; python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); r=d['roofline']; print('cfg3', d['value'], d['ms_per_step'], r['frac'], r['frac_algorithmic_floor'], r['traffic'], r['rows_decided_by_exact_kernel'], d['parity_gate'])"

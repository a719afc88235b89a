#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03v; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1 || { grep -E "^E|FAILED" $O/pytest_all.txt | head -20; }
tail -2 $O/pytest_all.txt
timeout -k 10 300 python scripts/dev/shard_sweep.py cfg2 2>&1 | grep -v amdgpu > $O/shard_sweep_cfg2.txt; cat $O/shard_sweep_cfg2.txt
timeout -k 10 300 python scripts/dev/shard_sweep.py cfg3 2>&1 | grep -v amdgpu > $O/shard_sweep_cfg3.txt; cat $O/shard_sweep_cfg3.txt

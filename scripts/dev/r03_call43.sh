#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03zz; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || { tail -30 $O/pytest_gpu.txt; exit 1; }
tail -2 $O/pytest_gpu.txt
timeout -k 10 400 python scripts/dev/soak.py 200 2>&1 | grep -v amdgpu | tee $O/soak.txt
bash scripts/dev/evidence.sh r03zz > $O/evidence.log 2>&1 || { tail -20 $O/evidence.log; exit 1; }
cut -c1-300 $O/bench.json

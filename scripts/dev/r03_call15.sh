#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03y; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1 || { grep -E "^E|FAILED" $O/pytest_all.txt | head -20; }
tail -2 $O/pytest_all.txt
export LIPVQ_SCREEN_MONITOR=0
LIPVQ_SCREEN_MODE=coarse timeout -k 10 600 python -m pytest tests/test_gpu_screen.py tests/test_gpu_random_shapes.py tests/test_gpu_fused.py -q -m gpu 2>&1 | tail -3
timeout -k 10 600 python scripts/dev/coarse_sweep.py 2>&1 | grep -v amdgpu > $O/coarse_sweep.txt; cat $O/coarse_sweep.txt

#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_screen.py tests/test_gpu_module.py tests/test_gpu_icl.py tests/test_gpu_random_shapes.py -x -q > $O/test_small.txt 2>&1 || { tail -40 $O/test_small.txt; exit 1; }
tail -3 $O/test_small.txt
timeout -k 10 300 python tests/bench_train_step.py 2>&1 | grep -v amdgpu | tee $O/bench_train_step2.txt

#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 500 python scripts/dev/soak_train.py 300 2>&1 | grep -v amdgpu | tee $O/soak_train.txt

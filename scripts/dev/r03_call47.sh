#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_big_parity.py tests/test_gpu_module.py -x -q > $O/test_asm.txt 2>&1 || { tail -30 $O/test_asm.txt; exit 1; }
tail -2 $O/test_asm.txt
for rep in 1 2 3; do
  for m in coarse fine; do
    r=$(LIPVQ_SCREEN_MODE=$m LIPVQ_SCREEN_MONITOR=0 timeout -k 10 200 python bench.py --workload icrt --metric-only --no-cpu-baseline --sustained 0 --steps 50 --warmup 20 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "icrt $m $r" | tee -a $O/slab_dma_asm.txt
  done
done
timeout -k 10 300 python scripts/dev/measure_train_big.py llfq icrt 2>&1 | grep "train step" | tee -a $O/slab_dma_asm.txt

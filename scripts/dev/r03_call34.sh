#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r03z; mkdir -p $O
for v in base pf16 pf32 base pf16 pf32; do
  lib=build_ab/$v/_lipvq_hip.so; [ $v = base ] && lib=lipvq-vae_amd/_lipvq_hip.so
  echo "== $v" | tee -a $O/pf_ab.txt
  LIPVQ_HIP_LIBRARY=$lib timeout -k 10 300 python tests/bench_train_step.py 2>&1 | grep "N=" | tee -a $O/pf_ab.txt
done
for v in base pf32; do
  lib=build_ab/$v/_lipvq_hip.so; [ $v = base ] && lib=lipvq-vae_amd/_lipvq_hip.so
  echo "== $v mid sizes" | tee -a $O/pf_ab.txt
  LIPVQ_HIP_LIBRARY=$lib timeout -k 10 300 python scripts/dev/measure_mlp3_mid.py 2>&1 | grep -v amdgpu | tee -a $O/pf_ab.txt
done

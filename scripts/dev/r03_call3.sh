#!/bin/bash
# GPU box, round 3 third call: VQVAE screened route (tests + timing), cfg3 ablations of tokenize_kernel<8>
set -e -o pipefail
cd "$(dirname "$0")/../.."
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_screen.py tests/test_gpu_module.py tests/test_gpu_comm.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
timeout -k 10 300 python scripts/dev/measure_vqvae.py > $O/vqvae.txt 2>&1; grep -v amdgpu $O/vqvae.txt
for v in st_base st_rg2 st_base st_rg2; do
  echo "== $v" >> $O/stamps_rg2.txt
  LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so timeout -k 10 200 python scripts/stamps.py cfg2 2>&1 | grep -v amdgpu >> $O/stamps_rg2.txt
done
cat $O/stamps_rg2.txt
BA="--metric-only --no-cpu-baseline --sustained 0 --traffic off --steps 30 --warmup 10"
for v in c3a_base c3a_noldsb c3a_nobar c3a_nostage c3a_notrack c3a_bare c3a_base; do
  echo "== $v" >> $O/cfg3_ablation.txt
  LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so timeout -k 10 200 python bench.py --workload cfg3 $BA 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 ms_per_step %.4f launch %.4f' % (d['ms_per_step'], d['roofline']['ms_per_launch']))" >> $O/cfg3_ablation.txt
done
cat $O/cfg3_ablation.txt

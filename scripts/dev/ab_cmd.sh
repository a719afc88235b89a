python -m pytest tests/test_gpu_fused.py tests/test_gpu_screen.py tests/test_gpu_random_shapes.py tests/test_gpu_big_parity.py tests/test_gpu_module.py tests/test_gpu_fast.py -x -q 2>&1 | tail -3
LIPVQ_HIP_LIBRARY=build_ab/rst/_lipvq_hip.so python scripts/measure_fused.py cfg2 2 2>&1 | grep "^x " | tail -2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in cfg2 cfg3 icrt; do bash scripts/prof_stats.sh s2k $wl 2>&1 | grep -E "nearest_rows" | cut -c1-40,150-260; done
for wl in cfg2 cfg2; do
  LIPVQ_HIP_LIBRARY=build_ab/base/_lipvq_hip.so python scripts/measure_fused.py $wl 300 2>&1 | grep -v amdgpu.ids
  python scripts/measure_fused.py $wl 300 2>&1 | grep -v amdgpu.ids
done

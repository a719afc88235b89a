python -m pytest tests/test_gpu_fused.py tests/test_gpu_screen.py tests/test_gpu_random_shapes.py tests/test_gpu_big_parity.py tests/test_gpu_module.py tests/test_gpu_fast.py tests/test_gpu_icl.py -x -q 2>&1 | tail -3
for wl in cfg2 cfg3 icrt; do
  echo "== $wl base"; LIPVQ_HIP_LIBRARY=build_ab/base/_lipvq_hip.so python scripts/measure_fused.py $wl 300 2>&1 | grep -v amdgpu.ids
  echo "== $wl new";  python scripts/measure_fused.py $wl 300 2>&1 | grep -v amdgpu.ids
done

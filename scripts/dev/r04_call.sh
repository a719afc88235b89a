set -e
mkdir -p gpurun_out/r04v
python -m pytest tests/test_gpu_fused.py tests/test_gpu_screen.py tests/test_gpu_module.py tests/test_gpu_big_parity.py -x -q > gpurun_out/r04v/pytest.txt 2>&1 || { tail -40 gpurun_out/r04v/pytest.txt; exit 1; }
tail -2 gpurun_out/r04v/pytest.txt
bash scripts/dev/ab_sweep.sh r04v cfg3 1,8 2 base main
bash scripts/dev/ab_sweep.sh r04v icrt 1,8 2 base main
bash scripts/dev/ab_sweep.sh r04v cfg2 1 2 base main

set -e
mkdir -p gpurun_out/r04z5
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off --metric-only > gpurun_out/r04z5/bench_plain_$rep.json 2> gpurun_out/r04z5/err.txt
LIPVQ_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off --metric-only > gpurun_out/r04z5/bench_nccl_world1_$rep.json 2> gpurun_out/r04z5/bench_nccl_world1.err
LIPVQ_BENCH_FORCE_DIST=1 LIPVQ_BENCH_COLLECTIVE=capi timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off --metric-only > gpurun_out/r04z5/bench_nccl_world1_capi_$rep.json 2>> gpurun_out/r04z5/bench_nccl_world1.err
LIPVQ_BENCH_FORCE_DIST=1 LIPVQ_BENCH_USAGE_BUCKET=16 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off --metric-only > gpurun_out/r04z5/bench_nccl_world1_m16_$rep.json 2>> gpurun_out/r04z5/bench_nccl_world1.err
done
for f in gpurun_out/r04z5/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], 'lines', sum(1 for _ in open('$f')), 'ms_per_step %.4f sustained %.4f ms_per_launch %.4f' % (d['ms_per_step'], d['sustained']['ms_per_step'], d['roofline']['ms_per_launch']))"; done

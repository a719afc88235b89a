#!/bin/bash
# GPU box, round 3 second call: cfg3 ring variants (same box A/B), e2-term ablation, kernel trace of a 65 536-row shard
set -e -o pipefail
cd "$(dirname "$0")/../.."
O=gpurun_out/r03b; mkdir -p $O
BA="--metric-only --no-cpu-baseline --sustained 0 --traffic off --steps 50 --warmup 20"
for v in c3_base c3_stream c3_s_tc2 c3_s_nb4 c3_base; do
  echo "== $v" >> $O/cfg3_ring_ab.txt
  LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so timeout -k 10 200 python bench.py --workload cfg3 $BA 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 ms_per_step %.4f launch %.4f exact_rows %s' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['rows_decided_by_exact_kernel']))" >> $O/cfg3_ring_ab.txt
done
cat $O/cfg3_ring_ab.txt
for v in c3_s_tc2 c3_s_nb4; do
  LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so timeout -k 10 400 python -m pytest tests/test_gpu_big_parity.py tests/test_gpu_random_shapes.py tests/test_gpu_fused.py -x -q -m gpu > $O/pytest_$v.txt 2>&1 || { tail -20 $O/pytest_$v.txt; }
  tail -2 $O/pytest_$v.txt
done
for v in abl_certall abl_noe2 abl_certall abl_noe2; do
  for wl in cfg2 cfg3; do
    echo "== $v $wl" >> $O/e2_ablation.txt
    LIPVQ_HIP_LIBRARY=build_ab/$v/_lipvq_hip.so timeout -k 10 200 python bench.py --workload $wl $BA 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f launch %.4f' % (d['ms_per_step'], d['roofline']['ms_per_launch']))" >> $O/e2_ablation.txt
  done
done
cat $O/e2_ablation.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace65k -- python3 scripts/dev/prof_shard.py cfg2 65536 300 > $O/prof_shard_65536.txt 2>$O/prof_shard.err
cat $O/prof_shard_65536.txt
cp $(ls $O/trace65k/*/*kernel_stats.csv | head -1) $O/kernel_stats_shard65536.csv
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace65k/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-900:]
import collections
gaps = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    ka, kb = a["Kernel_Name"].split("(")[0][:40], b["Kernel_Name"].split("(")[0][:40]
    gaps[ka + " -> " + kb].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
    dur[ka].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
out = open(sys.argv[1] + "/shard65536_timeline.txt", "w")
for k, v in dur.items(): print(f"duration {k:45s} n={len(v):4d} median {sorted(v)[len(v)//2]:8.2f} us", file=out)
for k, v in gaps.items(): print(f"gap      {k:85s} n={len(v):4d} median {sorted(v)[len(v)//2]:8.2f} us", file=out)
out.close()
print(open(sys.argv[1] + "/shard65536_timeline.txt").read())
PY
rm -rf $O/trace65k

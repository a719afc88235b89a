"""Dev soak (GPU): random (N, K, D, code distribution) cases of the counting-sort scatter routes against float64 index_add_
(accuracy), the scanning kernel (bit equality of the sequential route) and themselves (reproducibility).
python scripts/dev/soak_scatter.py [cases] [seed]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from lipvq_vae_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
gen = torch.Generator(device="cuda").manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), device="cuda", generator=gen).item())
bad = 0
for c in range(cases):
    N = ri(32768, 300000) if c % 4 else [32768, 65536, 8192 * 5, 2048 * 33 + 1][c // 4 % 4]
    K = [2, 3, 37, 256, 1000, 1024, 2048, 2049, 5000, 8192, 16384][ri(0, 10)]
    D = [1, 7, 16, 63, 64, 65, 128, 208][ri(0, 7)]
    kind = ri(0, 3)
    if kind == 0:
        idx = torch.randint(0, K, (N,), device="cuda", generator=gen)
    elif kind == 1:                                            # one code takes everything
        idx = torch.full((N,), ri(0, K - 1), device="cuda", dtype=torch.int64)
    elif kind == 2:                                            # heavy skew, many empty codes
        w = torch.rand(K, device="cuda", generator=gen) ** 12
        idx = torch.multinomial(w / w.sum(), N, replacement=True, generator=gen)
    else:                                                      # sorted runs (rows of a code adjacent)
        idx = torch.sort(torch.randint(0, K, (N,), device="cuda", generator=gen)).values
    g = torch.randn(N, D, device="cuda", generator=gen)
    ref = torch.zeros(K, D, device="cuda", dtype=torch.float64).index_add_(0, idx, g.double())
    a = ops.scatter_add(g, idx, K, route="sorted")
    b = ops.scatter_add(g, idx, K, route="sorted")
    s1 = ops.scatter_add(g, idx, K, route="sequential_sorted")
    s2 = ops.scatter_add(g, idx, K, route="sequential_scan")
    cnt = torch.bincount(idx, minlength=K)
    scale = max(float(ref.abs().max()), float(g.abs().max()) * float(cnt.max()) ** 0.5)
    ok = torch.equal(a, b) and torch.equal(s1, s2) and float((a.double() - ref).abs().max()) <= 3e-6 * scale \
        and float((s1.double() - ref).abs().max()) <= 2e-5 * scale and bool(torch.all(a[cnt == 0] == 0))
    if not ok:
        bad += 1
        print("FAIL", dict(N=N, K=K, D=D, kind=kind), torch.equal(a, b), torch.equal(s1, s2), float((a.double() - ref).abs().max()) / scale,
              float((s1.double() - ref).abs().max()) / scale)
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)

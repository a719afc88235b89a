import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from lipvq_vae_amd import ops
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for N, K, D in ((524288, 1024, 64), (524288, 1024, 208), (65536, 1024, 64)):
    g = torch.randn(N, D, device="cuda"); idx = torch.randint(0, K, (N,), device="cuda")
    w = torch.rand(K, device="cuda") ** 8
    idx2 = torch.multinomial(w / w.sum(), N, replacement=True)
    print(N, K, D, "det uniform %.1f us, det skewed %.1f us (max count %d); sorted uniform %.1f, skewed %.1f" % (
        timed(lambda: ops.scatter_add(g, idx, K, deterministic=True)), timed(lambda: ops.scatter_add(g, idx2, K, deterministic=True)),
        int(torch.bincount(idx2).max()), timed(lambda: ops.scatter_add(g, idx, K, route="sorted")), timed(lambda: ops.scatter_add(g, idx2, K, route="sorted"))), "scan-kernel det uniform %.1f us" % timed(lambda: ops.scatter_add(g, idx, K, route="sequential_scan")))

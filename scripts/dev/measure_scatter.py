import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from lipvq_vae_amd import ops
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for N, K, D in ((524288, 1024, 64), (524288, 8192, 128), (524288, 1024, 208), (65536, 1024, 64)):
    g = torch.randn(N, D, device="cuda"); idx = torch.randint(0, K, (N,), device="cuda")
    ref = torch.zeros(K, D, device="cuda", dtype=torch.float64).index_add_(0, idx, g.double())
    out = ops.scatter_add(g, idx, K)
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"scatter_add N={N} K={K} D={D}: {timed(lambda: ops.scatter_add(g, idx, K)):.1f} us (incl. the zero fill), rel err {err:.1e}")
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
cb = torch.rand(1024, 64, device="cuda")
print(f"nearest_prepare K=1024 D=64: {timed(lambda: ops.nearest_prepare(cb)):.1f} us")
cb = torch.rand(8192, 128, device="cuda")
print(f"nearest_prepare K=8192 D=128: {timed(lambda: ops.nearest_prepare(cb)):.1f} us")

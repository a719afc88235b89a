"""Dev measurement (GPU): backward of the embedding stage (embed_rows_bwd_kernel) at BASELINE config 2's batch, for spread and
for collapsed code indices (its table / time-embedding gradients are accumulated with fp32 atomics)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from lipvq_vae_amd import ops

B, T, E, K = 52428, 10, 512, 1024
N = B * T
dev = "cuda"
table = torch.randn(K, E, device=dev)
pos = torch.randn(T, E, device=dev) * 0.1
w, b = torch.ones(E, device=dev), torch.zeros(E, device=dev)
out = torch.empty(B, 3 * T, E, device=dev)
gout = torch.randn(B, 3 * T, E, device=dev)
for name, idx in (("uniform codes", torch.randint(0, K, (N,), device=dev)), ("8 codes", torch.randint(0, 8, (N,), device=dev)),
                  ("one code", torch.zeros(N, dtype=torch.int64, device=dev))):
    st = ops.embed_rows(table, idx, pos, w, b, 1e-5, out, N, T, 3 * T * E, 2 * E, E, want_stats=True)
    g_src, g_pos, g_w, g_b = torch.zeros_like(table), torch.zeros_like(pos), torch.zeros(E, device=dev), torch.zeros(E, device=dev)

    def run():
        ops.embed_rows_bwd(gout, table, idx, pos, st, w, g_src, g_pos, g_w, g_b, N, T, 3 * T * E, 2 * E, E)
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print(f"embed_rows_bwd N={N} E={E} K={K} T={T}, {name}: {e0.elapsed_time(e1) / 5:.3f} ms")

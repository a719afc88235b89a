"""Dev tool (GPU): where the exact-rows route (lipvq_nearest_rows_f32) stops beating prepare + screen."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from lipvq_vae_amd import ops


def timed(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for K, D in ((1024, 208), (1024, 64), (256, 32)):
    cb = torch.rand(K, D, device="cuda")
    prep = ops.nearest_prepare(cb)
    t_prep = timed(lambda: ops.nearest_prepare(cb))
    for N in (80, 500, 2048, 4096, 8192, 16384, 65536):
        z = torch.rand(N, D, device="cuda")
        t_rows = timed(lambda: ops.nearest_rows(z, cb))
        t_scr = timed(lambda: ops.nearest_screened(z, cb, prep))
        print(f"K={K} D={D} N={N}: rows {t_rows:.1f} us | screen {t_scr:.1f} us (+ prepare {t_prep:.1f} us when the codebook changed)")

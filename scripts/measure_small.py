"""Dev tool (GPU): latency of the small-batch (training-step sized) launches, kernel by kernel."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from lipvq_vae_amd import ops
from lipvq_vae_amd.ops import ACT_GELU, ACT_NONE, ACT_SIGMOID, ACT_RELU


def timed(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


import os
print("LIPVQ_MLP3_SMALL_TILES =", os.environ.get("LIPVQ_MLP3_SMALL_TILES", "(default)"))
for (N, K0, J0, J1, J2) in [(80, 12, 64, 128, 208), (80, 208, 64, 128, 12), (500, 12, 64, 128, 208), (2048, 12, 64, 128, 208),
                            (8192, 12, 64, 128, 208), (16384, 7, 64, 128, 64), (32768, 7, 64, 128, 64), (131072, 7, 64, 128, 64),
                            (524288, 7, 64, 128, 64), (131072, 12, 64, 128, 208)]:
    Ws = [torch.randn(J0, K0, device="cuda") / K0 ** 0.5, torch.randn(J0, device="cuda"), torch.randn(J1, J0, device="cuda") / 8,
          torch.randn(J1, device="cuda"), torch.randn(J2, J1, device="cuda") / 11, torch.randn(J2, device="cuda")]
    packed = ops.mlp3_pack(*Ws)
    x = torch.randn(N, K0, device="cuda")
    for acts, name in (((ACT_GELU, ACT_GELU, ACT_SIGMOID), "gelu,gelu,sigmoid"),):
        t = timed(lambda: ops.mlp3(x, packed, acts))
        t2 = timed(lambda: ops.mlp3(x, packed, acts, save_pre=True))
        print(f"mlp3 N={N} {K0}->{J0}->{J1}->{J2} acts={name}: {t:.1f} us (save_pre {t2:.1f} us)")
    pb = ops.mlp3_pack_bwd(Ws[0], Ws[2], Ws[4])
    y, pre = ops.mlp3(x, packed, (ACT_GELU, ACT_GELU, ACT_SIGMOID), save_pre=True)
    gy = torch.randn_like(y)
    tb = timed(lambda: ops.mlp3_bwd(gy, pre, pb, (ACT_GELU, ACT_GELU, ACT_SIGMOID), want_gx=False))
    print(f"mlp3_bwd N={N}: {tb:.1f} us")

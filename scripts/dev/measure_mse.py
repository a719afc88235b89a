"""Dev measurement (GPU): the loss launch (two mse means + loss) at BASELINE config 2's batch."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from lipvq_vae_amd import ops
N, A, D = 524288, 7, 64
xr, x = torch.randn(N, A, device="cuda"), torch.randn(N, A, device="cuda")
zq, ze = torch.rand(N, D, device="cuda"), torch.rand(N, D, device="cuda")
for _ in range(3): ops.mse_pair_loss(xr, x, zq, ze, 0.25, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.mse_pair_loss(xr, x, zq, ze, 0.25, 0)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"mse_pair_loss N={N}: {us:.1f} us, {4.0 * N * 2 * (A + D) / us / 1e3:.0f} GB/s")

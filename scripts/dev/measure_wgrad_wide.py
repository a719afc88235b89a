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
for N, J, Kd in ((524280, 512, 64), (524280, 256, 64), (163840, 512, 64), (524280, 512, 128)):
    G, H = torch.randn(N, J, device="cuda"), torch.randn(N, Kd, device="cuda")
    t = timed(lambda: ops.wgrad(G, H))
    print(f"wgrad N={N} J={J} Kd={Kd}: {t:.1f} us, {2.0 * N * J * Kd / t / 1e6:.1f} TFLOP/s, {4.0 * N * (J + Kd) / t / 1e3:.0f} GB/s")

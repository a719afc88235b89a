"""Dev tool (GPU): wgrad at BASELINE config 2's batch for the layer shapes of the tokenizer (LIPVQ_WGRAD_CHUNK overrides
the rows per chunk)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
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


print("LIPVQ_WGRAD_CHUNK =", os.environ.get("LIPVQ_WGRAD_CHUNK", "(default)"))
for N in (524288, 65536):
    for J, Kd, act in ((128, 64, ops.ACT_GELU), (64, 128, ops.ACT_GELU), (64, 7, ops.ACT_NONE), (7, 128, ops.ACT_GELU), (64, 64, ops.ACT_NONE),
                       (128, 64, ops.ACT_NONE)):
        G, H = torch.randn(N, J, device="cuda"), torch.randn(N, Kd, device="cuda")
        t = timed(lambda: ops.wgrad(G, H, h_act=act))
        print(f"wgrad N={N} J={J} Kd={Kd} act={act}: {t:.1f} us, {2.0 * N * J * Kd / t / 1e6:.1f} TFLOP/s, {4.0 * N * (J + Kd) / t / 1e3:.0f} GB/s")

"""Dev tool (GPU): host-side cost of issuing the small launches of a training step (Python wrapper + ctypes + HIP)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from lipvq_vae_amd import ops


def per_call(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    dt_issue = time.perf_counter() - t
    torch.cuda.synchronize(); dt_all = time.perf_counter() - t
    return dt_issue / n * 1e6, dt_all / n * 1e6


a, b = torch.randn(80, 208, device="cuda"), torch.randn(80, 208, device="cuda")
dev = a.device
print("ops.ste (1 tiny kernel): issue %.1f us, incl. GPU %.1f us" % per_call(lambda: ops.ste(a, b)))
print("torch add (1 tiny kernel): issue %.1f us, incl. GPU %.1f us" % per_call(lambda: torch.add(a, b)))
print("torch.empty_like: %.1f us" % per_call(lambda: torch.empty_like(a))[0])
def ctx():
    with torch.cuda.device(dev):
        pass
print("with torch.cuda.device(dev): %.1f us" % per_call(ctx)[0])
print("torch.cuda.current_stream().cuda_stream: %.1f us" % per_call(lambda: torch.cuda.current_stream().cuda_stream)[0])
print("ops._chk x2: %.1f us" % per_call(lambda: (ops._chk(a, "a"), ops._chk(b, "b")))[0])

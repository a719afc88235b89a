"""Dev measurement (GPU): the bin tokenizer (AdaptiveBinActionEmbedding) forward + backward at BASELINE config 2's batch."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.binning import AdaptiveBinActionEmbedding

N, A, D = 524288, 7, 64
m = AdaptiveBinActionEmbedding(A, D).cuda()
x = torch.randn(N, A, device="cuda")
m(x)
m._update_enabled = False


def step():
    m.zero_grad()
    m(x).square().mean().backward()


def fwd():
    with torch.no_grad():
        m(x)


for name, fn in (("forward", fwd), ("forward + backward", step)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"bin tokenizer N={N} A={A} D={D}: {name} {e0.elapsed_time(e1) / 5:.3f} ms")

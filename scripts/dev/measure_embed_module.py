"""Dev measurement (GPU): ICLInputEmbedding forward + backward (three streams into [B, 3T, E]) at a large batch."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.embedding import ICLInputEmbedding

B, T, Din, E, K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 10, 64, 512, 1024
m = ICLInputEmbedding(Din, E, T, emb_dropout=0.0).cuda()
obs, cobs = torch.randn(B, T, Din, device="cuda"), torch.randn(B, T, Din, device="cuda")
idx = torch.randint(0, K, (B, T), device="cuda")
cb = torch.rand(K, Din, device="cuda")


def fwd():
    with torch.no_grad():
        m(obs, cobs, action_indices=idx, codebook=cb)


def step():
    m.zero_grad()
    m(obs, cobs, action_indices=idx, codebook=cb).square().mean().backward()


for name, fn in (("forward", fwd), ("forward + backward", step)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"ICLInputEmbedding B={B} T={T} E={E} ({B * T} actions): {name} {e0.elapsed_time(e1) / 5:.3f} ms")

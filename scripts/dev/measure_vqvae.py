"""Dev measurement (GPU): the plain VQVAE (reference backbone.py) at BASELINE config 2's batch: tokenize, full forward, training step."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.tokenizer import VQVAE

N, A, D = 524288, 7, 64
for K in (128, 1024):
    torch.manual_seed(0)
    m = VQVAE(A, D, num_embeddings=K).cuda()
    with torch.no_grad():
        m.embedding.weight.copy_(torch.rand(K, D, device="cuda"))
    x = torch.randn(N, A, device="cuda")
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)

    def timed(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    def fwd():
        with torch.no_grad():
            m(x)

    def step():
        opt.zero_grad()
        _, loss = m(x)
        loss.backward()
        opt.step()

    t_tok, t_fwd, t_step = timed(lambda: m.tokenize(x)), timed(fwd), timed(step, 5)
    m.tokenize(x)
    ex = int(m.last_exact_rows[0]) if m.last_exact_rows is not None else None
    print(f"VQVAE N={N} A={A} D={D} K={K}: tokenize {t_tok:.3f} ms ({N / t_tok / 1e3:.0f} M actions/s), full forward {t_fwd:.3f} ms, "
          f"training step {t_step:.3f} ms; rows decided by the exact kernel after the screen: {ex}")

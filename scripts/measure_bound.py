"""Dev measurement (GPU): largest observed |d~ - d| relative to the certification bound's scale, over many pairs."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import lipvq_vae_amd
from lipvq_vae_amd import ops
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import trained_like_
G = 2.0 ** -18
worst = 0.0
for (A, D, K, N, seed) in [(7, 64, 1024, 4096, 0), (7, 64, 1024, 4096, 1), (7, 128, 8192, 1024, 2), (7, 32, 256, 8192, 3), (12, 208, 1024, 512, 4)]:
    torch.manual_seed(seed)
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    trained_like_(model, A, seed=seed)
    x = torch.randn(N, A, device="cuda")
    z = model.encode(x)
    cb = model.quantizer.codebook.detach()
    prep = ops.nearest_prepare(cb)
    _, _, dt = ops.nearest_screened(z, cb, prep, debug_gamma=G)
    dt = dt[:, :K].double()
    mu = cb.double().mean(0)
    zc, ec = z.double() - mu, cb.double() - mu
    d = (ec * ec).sum(1)[None, :] - 2.0 * zc @ ec.T
    e2max = (ec * ec).sum(1).max()
    scale = (e2max + 2.0 * (zc * zc).sum(1).sqrt() * e2max.sqrt())[:, None]
    ratio = ((dt - d).abs() / scale).max().item()
    print(f"A={A} D={D} K={K} N={N}: max |d~-d| / (E2max + 2|z'|Emax) = {ratio:.3e} = 2^{np.log2(ratio):.1f}   ({N*K:.1e} pairs)")
    worst = max(worst, ratio)
print(f"worst = 2^{np.log2(worst):.2f}; gamma = 2^-18 leaves a factor {G/worst:.1f}")

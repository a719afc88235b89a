"""Dev check (GPU): one tokenize launch over a batch whose z_q exceeds 2^31 bytes equals the same rows tokenized in pieces
(indices, z_q, usage); also a training step's gradients at that size are finite.  python scripts/dev/check_huge_batch.py [rows]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import trained_like_

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_003
A, D, K = 7, 64, 1024
torch.manual_seed(0)
m = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(m, A)
x = torch.randn(N, A, device="cuda")
m.reset_usage()
idx, zq = m.tokenize(x)
u_all = m.code_usage.clone()
m.reset_usage()
ok = True
for a in range(0, N, 3_000_001):
    b = min(N, a + 3_000_001)
    i2, z2 = m.tokenize(x[a:b].contiguous())
    ok = ok and torch.equal(i2, idx[a:b]) and torch.equal(z2, zq[a:b])
ok = ok and torch.equal(m.code_usage, u_all) and int(u_all.sum()) == N
print(f"N={N}: one launch == pieces: {ok}; z_q bytes {zq.numel() * 4}")
del zq, idx
_, loss = m(x[:6_000_000])
loss.backward()
fin = all(torch.isfinite(p.grad).all().item() for p in m.parameters())
print(f"training forward+backward at 6 000 000 rows: loss {float(loss):.6f}, gradients finite: {fin}")
sys.exit(0 if ok and fin else 1)

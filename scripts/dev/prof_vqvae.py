import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch, lipvq_vae_amd
from lipvq_vae_amd.tokenizer import VQVAE
N, A, D = 524288, 7, 64
for K in (128, 1024):
    m = VQVAE(A, D, num_embeddings=K).cuda()
    with torch.no_grad(): m.embedding.weight.copy_(torch.rand(K, D, device="cuda"))
    x = torch.randn(N, A, device="cuda")
    for _ in range(5): m.tokenize(x)
torch.cuda.synchronize()

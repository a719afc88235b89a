import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from lipvq_vae_amd.icl import GraphedTokenizerStep
from bench import trained_like_
N, A, D, K = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (80, 12, 208, 1024)
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
x = torch.randn(N, A, device="cuda")
g = GraphedTokenizerStep(model, x)
for _ in range(50):
    g.step(x)
torch.cuda.synchronize()

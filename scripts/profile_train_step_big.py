"""Dev tool (GPU): kernel breakdown of a training step at BASELINE config 2's batch (524 288 rows); run under rocprofv3."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import trained_like_
from lipvq_vae_amd.optim import AdamW

N, A, D, K = 524288, 7, 64, 1024
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
opt = (torch.optim.AdamW if os.environ.get('LIPVQ_TORCH_ADAMW') == '1' else AdamW)(model.parameters(), lr=1e-3, weight_decay=1e-4)
x = torch.randn(N, A, device="cuda")
for _ in range(6):
    opt.zero_grad()
    _, loss = model(x)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("done")

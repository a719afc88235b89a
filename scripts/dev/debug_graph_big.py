"""Dev tool (GPU): does GraphedTokenizerStep capture a large-batch step (screened quantizer, folded training routes)?"""
import sys, copy
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from lipvq_vae_amd.icl import GraphedTokenizerStep, VQTokenizerTrainer
from bench import trained_like_
N, A, D, K = int(sys.argv[1]) if len(sys.argv) > 1 else 66000, 7, 64, 1024
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda(); trained_like_(model, A)
twin = copy.deepcopy(model); twin.invalidate_caches()
tr, tw = VQTokenizerTrainer(model), VQTokenizerTrainer(twin)
xs = [torch.randn(N, A, device="cuda") for _ in range(5)]
g = GraphedTokenizerStep(model, xs[0], optimizer_state=tr.vq_optimizer.state_dict(), warmup=2)
for i in range(1, 4):
    _, loss = g.step(xs[i]); _, ref = tw.train_on_actions(xs[i])
    print(i, float(loss), float(ref))
torch.cuda.synchronize()
print("max param diff", max(float((a - b).abs().max()) for a, b in zip(model.state_dict().values(), twin.state_dict().values())))

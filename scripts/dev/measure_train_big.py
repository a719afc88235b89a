"""Dev tool (GPU): the training step (zero_grad + forward + backward + AdamW) at BASELINE config 2's batch, warmed, HIP events;
with LIPVQ_NO_FOLD=1 the loss-gradient terms go through the separate lipvq_scaled_diff_f32 launches (round-2 route)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd import ops
from lipvq_vae_amd.tokenizer import LLFQVAE_V4, VQVAE
from bench import trained_like_
from lipvq_vae_amd.optim import AdamW

kind = sys.argv[1] if len(sys.argv) > 1 else "llfq"
N, A, D, K = {"cfg3": (524288, 7, 128, 8192), "icrt": (524280, 12, 208, 1024)}.get(sys.argv[2] if len(sys.argv) > 2 else "", (524288, 7, 64, 1024))
if os.environ.get("LIPVQ_NO_FOLD") == "1":
    ops.mlp3_bwd_vq_supported = lambda N, pk: False
torch.manual_seed(0)
if kind == "llfq":
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    trained_like_(model, A)
else:
    model = VQVAE(A, D, num_embeddings=int(os.environ.get("VQ_K", "128"))).cuda()
    if os.environ.get("LIPVQ_VQ_TRAIN_UNFUSED") == "1":
        VQVAE.fused_shape = lambda self: False
opt = (torch.optim.AdamW if os.environ.get('LIPVQ_TORCH_ADAMW') == '1' else AdamW)(model.parameters(), lr=1e-3, weight_decay=1e-4)
x = torch.randn(N, A, device="cuda")
def step():
    opt.zero_grad()
    _, loss = model(x)
    loss.backward()
    opt.step()
for _ in range(15): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = []
for rep in range(3):
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 20)
print(f"{kind} train step N={N} A={A} D={D} K={K if kind == 'llfq' else model.num_embeddings}: " + " / ".join(f"{b:.3f}" for b in best) + " ms" + ("  (separate scaled_diff launches)" if os.environ.get("LIPVQ_NO_FOLD") == "1" else "  (folded terms)")
      + (" unfused training forward" if kind != "llfq" and os.environ.get("LIPVQ_VQ_TRAIN_UNFUSED") == "1" else ""))

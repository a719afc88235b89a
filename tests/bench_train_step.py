"""Tokenizer training step (zero_grad -> forward -> loss.backward() -> AdamW.step(), icl.py:913-914,968-970) at the
real ICRT shapes: GPU (this library) vs the torch-CPU restatement of the reference with 1 thread
(what the reference's train() sets, scripts/train.py:57).  Prints ms/step."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))  # repo root
import torch
import lipvq_vae_amd
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from lipvq_vae_amd.icl import GraphedTokenizerStep, VQTokenizerTrainer
from bench import trained_like_
from oracle import lipvq_oracle as O

for (N, A, D, K) in [(80, 12, 208, 1024), (500, 12, 208, 1024), (1024, 7, 32, 256)]:
    torch.manual_seed(0)
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    trained_like_(model, A)
    tr = VQTokenizerTrainer(model)
    x = torch.randn(N, A, device="cuda")
    for _ in range(5): tr.train_on_actions(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 50
    for _ in range(n): tr.train_on_actions(x)
    torch.cuda.synchronize(); gpu_ms = (time.perf_counter() - t) / n * 1e3
    # the same step captured into a HIP graph (GraphedTokenizerStep): one host call per step
    gs = GraphedTokenizerStep(model, x, optimizer_state=tr.vq_optimizer.state_dict())
    for _ in range(5): gs.step(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): gs.step(x)
    torch.cuda.synchronize(); graph_ms = (time.perf_counter() - t) / n * 1e3
    torch.set_num_threads(1)
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    opt = torch.optim.AdamW(list(p.values()), lr=1e-3, weight_decay=1e-4)
    xc = x.cpu()
    def cpu_step():
        opt.zero_grad(); _, loss, _ = O.torch_llfq_forward(p, xc); loss.backward(); opt.step()
    cpu_step(); t = time.perf_counter(); m = 5
    for _ in range(m): cpu_step()
    cpu_ms = (time.perf_counter() - t) / m * 1e3
    print(f"N={N} A={A} D={D} K={K}: GPU eager {gpu_ms:.3f} ms/step, HIP-graph replay {graph_ms:.3f} ms/step, torch-CPU (1 thread) {cpu_ms:.1f} ms/step, ratio {cpu_ms/gpu_ms:.0f}x")

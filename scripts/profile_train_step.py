"""Dev tool (GPU): where a tokenizer training step at the ICRT shape spends its time -- wall clock per step vs the sum
of kernel durations (run under `rocprofv3 --kernel-trace --stats --output-format csv`)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from lipvq_vae_amd.icl import VQTokenizerTrainer
from bench import trained_like_

N, A, D, K = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (80, 12, 208, 1024)
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
tr = VQTokenizerTrainer(model)
x = torch.randn(N, A, device="cuda")
for _ in range(5):
    tr.train_on_actions(x)
torch.cuda.synchronize()
n = 100
t = time.perf_counter()
for _ in range(n):
    tr.train_on_actions(x)
torch.cuda.synchronize()
print(f"wall {1e3 * (time.perf_counter() - t) / n:.3f} ms/step over {n} steps")
# phases
def timed(fn, n=100):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / n
fwd = lambda: model(x)
print(f"forward only (grad on) {timed(fwd):.3f} ms")
def fb():
    _, loss = model(x); loss.backward()
print(f"forward+backward {timed(fb):.3f} ms")
opt = tr.vq_optimizer
print(f"AdamW.step alone {timed(opt.step):.3f} ms; zero_grad {timed(opt.zero_grad):.3f} ms")
with torch.no_grad():
    print(f"forward no-grad {timed(fwd):.3f} ms")

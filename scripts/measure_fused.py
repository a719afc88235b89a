"""Dev measurement (GPU): fused tokenize at a BASELINE shape, n launches (for rocprofv3 runs)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import lipvq_vae_amd
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import WORKLOADS, trained_like_
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, T, A, D, K = WORKLOADS[wl]
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
x = torch.randn(B * T, A, device="cuda")
model.tokenize(x); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
import os
cu = os.environ.get('LQ_NO_USAGE') is None
for _ in range(n): model.tokenize(x, count_usage=cu)
e1.record(); torch.cuda.synchronize()
print(f"{wl}: fused tokenize {e0.elapsed_time(e1)/n:.3f} ms/launch, rows to exact kernel {int(model.last_exact_rows[0])}")

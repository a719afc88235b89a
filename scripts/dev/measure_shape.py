"""Dev measurement (GPU): fused tokenize at an arbitrary shape: python scripts/dev/measure_shape.py A D K [N] [launches]"""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import trained_like_
A, D, K = (int(v) for v in sys.argv[1:4])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 524288
n = int(sys.argv[5]) if len(sys.argv) > 5 else 200
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
x = torch.randn(N, A, device="cuda")
cu = os.environ.get('LQ_NO_USAGE') is None
idx0, _ = model.tokenize(x, count_usage=cu); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): model.tokenize(x, count_usage=cu)
e1.record(); torch.cuda.synchronize()
print(f"A={A} D={D} K={K} N={N}: fused tokenize {e0.elapsed_time(e1)/n:.3f} ms/launch, rows to exact kernel {int(model.last_exact_rows[0])}, idx checksum {int(idx0.sum())}")

"""Dev driver for `rocprofv3 --kernel-trace --stats`: LLFQVAE_V4.tokenize at ROWS rows of a workload, LOOPS times.
   python scripts/dev/prof_shard.py [workload] [rows] [loops]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import WORKLOADS, trained_like_
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B, T, A, D, K = WORKLOADS[wl]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
loops = int(sys.argv[3]) if len(sys.argv) > 3 else 200
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
x = torch.randn(n, A, generator=torch.Generator(device="cpu").manual_seed(1234)).cuda()
for _ in range(50):
    model.tokenize(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(loops):
    model.tokenize(x)
e1.record()
torch.cuda.synchronize()
print(f"{wl} rows {n}: {e0.elapsed_time(e1) / loops * 1e3:.1f} us per tokenize call ({loops} calls)")

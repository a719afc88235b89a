"""Dev measurement (GPU box): the compute-side ceiling of BASELINE config 4 (cfg2's global batch sharded B/G per GPU)
on ONE GPU: LLFQVAE_V4.tokenize at the shard sizes of G = 1, 2, 4, 8, 16, 32 (524 288 ... 16 384 rows).
   python scripts/dev/shard_sweep.py [workload] [--fast]        (SWEEP_G=1,8: only those shard counts)"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import WORKLOADS, trained_like_

wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfg2"
mode = "fast" if "--fast" in sys.argv else "parity"
B, T, A, D, K = WORKLOADS[wl]
N = B * T
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
xfull = torch.randn(N, A, generator=torch.Generator(device="cpu").manual_seed(1234)).cuda()
idx_full, _ = model.tokenize(xfull, mode=mode)
print(f"{wl} ({mode}): A={A} D={D} K={K}; one GPU, shard = first N/G rows of the global batch")
print(f"{'G':>3} {'rows':>8} {'ms/launch':>10} {'M actions/s':>12} {'G x rate (ideal 1->G)':>22} {'exact rows':>10} {'== full-batch idx':>18}")
for G in tuple(int(g) for g in os.environ.get("SWEEP_G", "1,2,4,8,16,32").split(",")):
    n = N // G
    x = xfull[:n].contiguous()
    for _ in range(200):
        model.tokenize(x, mode=mode)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            idx, _ = model.tokenize(x, mode=mode)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 300)
    ex = int(model.last_exact_rows[0]) if model.last_exact_rows is not None else -1
    same = bool(torch.equal(idx, idx_full[:n]))
    print(f"{G:3d} {n:8d} {best:10.4f} {n / best / 1e3:12.1f} {G * n / best / 1e3:22.1f} {ex:10d} {str(same):>18}", flush=True)

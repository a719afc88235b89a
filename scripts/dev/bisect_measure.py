"""Dev measurement (GPU box): time tokenize (parity and fast mode) of ONE archived source tree (scripts/dev/bisect_trees.sh)
at the metric's shape, in a process of its own:  python scripts/dev/bisect_measure.py build_ab/trees/<rev> [rows]"""
import sys
from pathlib import Path
tree = Path(sys.argv[1]).resolve()
sys.path.insert(0, str(tree))
import torch
import lipvq_vae_amd  # noqa: F401  (the tree's own package and library)
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
assert str(tree) in lipvq_vae_amd.__file__, lipvq_vae_amd.__file__
A, D, K = 7, 64, 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096 * 128
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
g = torch.Generator(device="cpu").manual_seed(0)
with torch.no_grad():                      # bench.trained_like_, spelled out so that it does not depend on the tree's bench.py
    model.to_latent.ci.fill_(40.0)
    model.to_latent.b.copy_(torch.randn(D, generator=g).cuda())
    cb = torch.rand(K, D, generator=g).cuda()
    xs = torch.randn(4 * K, A, generator=g).cuda()
    ze = model.encode(xs)
    pick = torch.randperm(xs.shape[0], generator=g)[: K // 2].cuda()
    cb[: K // 2] = ze[pick] + 0.02 * torch.randn(K // 2, D, generator=g).cuda()
    model.quantizer.codebook.copy_(cb)
x = torch.randn(N, A, generator=torch.Generator(device="cpu").manual_seed(1234)).cuda()


def timed(fn, n=200):
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


par = timed(lambda: model.tokenize(x, count_usage=False))
idx_p, _ = model.tokenize(x, count_usage=False)
try:
    fast = timed(lambda: model.tokenize(x, count_usage=False, mode="fast"))
    idx_f, _ = model.tokenize(x, count_usage=False, mode="fast")
    flips = float((idx_f != idx_p).float().mean().item())
except Exception as e:  # noqa: BLE001
    fast, flips = float("nan"), float("nan")
print(f"{tree.name:10s} rows {N}: parity {par:.4f} ms  fast {fast:.4f} ms  flips {flips:.2e}  idx checksum {int(idx_p.sum().item())}", flush=True)

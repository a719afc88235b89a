"""Dev measurement (GPU): screened vs direct nearest at a BASELINE shape; prints times and the fraction
of rows that needed the exact kernel.  Usage: python scripts/measure_screen.py [cfg2|cfg3]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import lipvq_vae_amd
from lipvq_vae_amd import ops
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import WORKLOADS, trained_like_

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B, T, A, D, K = WORKLOADS[wl]
N = B * T
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
x = torch.randn(N, A, device="cuda")
z = model.encode(x)
cb = model.quantizer.codebook.detach()

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

t_prep = timeit(lambda: ops.nearest_prepare(cb))
prep = ops.nearest_prepare(cb)
idx_d, zq_d, _ = ops.nearest(z, cb)
idx_s, zq_s, ws = ops.nearest_screened(z, cb, prep, return_workspace=True)
print("equal idx:", torch.equal(idx_d, idx_s), "equal zq:", torch.equal(zq_d, zq_s), "rows to exact kernel:", int(ws[0]), f"({100.0*int(ws[0])/N:.3f}%)")
if wl == "cfg2" or "--direct" in sys.argv:
    print(f"direct   : {timeit(lambda: ops.nearest(z, cb), 5):8.3f} ms")
print(f"screened : {timeit(lambda: ops.nearest_screened(z, cb, prep)):8.3f} ms   (prepare {t_prep:.3f} ms)")
print(f"encode   : {timeit(lambda: model.encode(x)):8.3f} ms")
import os
cu = os.environ.get("LQ_NO_USAGE") is None
print(f"fused tokenize: {timeit(lambda: model.tokenize(x, count_usage=cu)):8.3f} ms  rows to exact: {int(model.last_exact_rows[0])}")

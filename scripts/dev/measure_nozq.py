import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd
from lipvq_vae_amd import ops
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import WORKLOADS, trained_like_
B, T, A, D, K = WORKLOADS["cfg2"]
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
x = torch.randn(B * T, A, device="cuda")
packed, _, Wn = model._packed_encoder()
w0, b0, w1, b1, _, b2, _ = (t.detach() for t in model._enc_params())
raw = (w0, b0, w1, b1, Wn, b2)
cb = model.quantizer.codebook.detach()
prep = ops.nearest_prepare(cb)
ws = ops.tokenize_workspace(B * T, D, x.device)
usage = torch.zeros(K, dtype=torch.int64, device="cuda")
def timed(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for name, kw in (("zq + usage", dict(usage=usage, want_zq=True)), ("zq, no usage", dict(usage=None, want_zq=True)),
                 ("no zq, usage", dict(usage=usage, want_zq=False)), ("no zq, no usage", dict(usage=None, want_zq=False)),
                 ("zq + usage + ze", dict(usage=usage, want_zq=True, want_ze=True))):
    print(f"{name:18s} {timed(lambda: ops.tokenize(x, packed, raw, cb, prep, workspace=ws, **kw)):.4f} ms")

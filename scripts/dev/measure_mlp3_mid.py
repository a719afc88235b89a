"""Dev tool (GPU): mlp3_wg_kernel forward / backward at batch sizes between the graphed step and the LDS-resident kernel."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd
from lipvq_vae_amd import ops
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from lipvq_vae_amd.autograd import _ENC_ACTS
from bench import trained_like_
for (N, A, D) in [(80, 12, 208), (500, 12, 208), (2048, 12, 208), (8192, 7, 64), (32768, 7, 64), (32768, 12, 208)]:
    torch.manual_seed(0)
    m = LLFQVAE_V4(A, D, num_codes=1024).cuda(); trained_like_(m, A)
    x = torch.randn(N, A, device="cuda")
    pk = m._packed_encoder()[0]
    def fwd(): return ops.mlp3(x, pk, _ENC_ACTS, save_pre=True)
    ze, pre = fwd()
    e = m.encoder
    pkb = ops.mlp3_pack_bwd(e[0].weight.detach(), e[2].weight.detach(), m._packed_encoder()[2])
    g = torch.randn_like(ze)
    def bwd(): return ops.mlp3_bwd(g, pre, pkb, _ENC_ACTS, want_gx=False)
    out = []
    for fn in (fwd, bwd):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 200 * 1e3)
    print(f"N={N} A={A} D={D}: encoder forward (saving) {out[0]:.1f} us, backward {out[1]:.1f} us (eager launches back to back)")

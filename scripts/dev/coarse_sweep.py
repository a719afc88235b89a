"""Dev measurement (GPU): LLFQVAE_V4.tokenize under the three-product and the one-product screen (lipvq_set_option("screen_mode"),
read per launch) over (D, K) -- the data behind lq_screen_coarse_default (lipvq_screen.h).   python scripts/dev/coarse_sweep.py"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd import _capi
from lipvq_vae_amd.tokenizer import LLFQVAE_V4, _ScreenMonitor
_ScreenMonitor.ENABLED = False
from bench import trained_like_

N = 4096 * 128
print(f"{N} rows; ms per tokenize call (fused launch + exact stage), rows left to the exact stage")
for A, D, Ks in ((7, 64, (1024, 4096, 8192)), (7, 128, (1024, 2048, 4096, 8192)), (12, 208, (1024, 4096, 8192))):
    for K in Ks:
        torch.manual_seed(0)
        model = LLFQVAE_V4(A, D, num_codes=K).cuda()
        trained_like_(model, A)
        x = torch.randn(N, A, generator=torch.Generator(device="cpu").manual_seed(1234)).cuda()
        out = {}
        ref = None
        for mode in ("fine", "coarse"):
            _capi.set_option("screen_mode", mode)
            for _ in range(30):
                idx, _ = model.tokenize(x, count_usage=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(60):
                idx, _ = model.tokenize(x, count_usage=False)
            e1.record()
            torch.cuda.synchronize()
            out[mode] = (e0.elapsed_time(e1) / 60, int(model.last_exact_rows[0]))
            ref = idx if ref is None else ref
            assert torch.equal(idx, ref), "the two screens disagree"
        _capi.set_option("screen_mode", None)
        print(f"A={A:2d} D={D:3d} K={K:5d}: fine {out['fine'][0]:7.3f} ms ({out['fine'][1]:6d} rows)   coarse {out['coarse'][0]:7.3f} ms "
              f"({out['coarse'][1]:6d} rows)   coarse/fine {out['coarse'][0] / out['fine'][0]:.2f}", flush=True)

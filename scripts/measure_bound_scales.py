"""Dev measurement (GPU): screening error relative to the bound's scale when all operands are scaled by s
(probes fp16 denormal handling of the lo pieces inside v_mfma_f32_32x32x16_f16)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import lipvq_vae_amd
from lipvq_vae_amd import ops
G = 2.0 ** -18
rng = np.random.default_rng(0)
K, D, N = 512, 64, 1024
cb0 = rng.uniform(0, 1, (K, D)).astype(np.float32); z0 = rng.uniform(0, 1, (N, D)).astype(np.float32)
for s in (1.0, 1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e2, 1e3, 1e4):
    cb = torch.from_numpy(cb0 * np.float32(s)).cuda(); z = torch.from_numpy(z0 * np.float32(s)).cuda()
    prep = ops.nearest_prepare(cb)
    idx, _, ws, dt = ops.nearest_screened(z, cb, prep, return_workspace=True, debug_gamma=G)
    idx_d, _, _ = ops.nearest(z, cb)
    dt = dt[:, :K].double()
    mu = cb.double().mean(0); zc, ec = z.double() - mu, cb.double() - mu
    d = (ec * ec).sum(1)[None, :] - 2.0 * zc @ ec.T
    e2max = (ec * ec).sum(1).max()
    scale = (e2max + 2.0 * (zc * zc).sum(1).sqrt() * e2max.sqrt())[:, None]
    ratio = ((dt - d).abs() / scale).max().item()
    print(f"scale {s:8.0e}: max err/scale = 2^{np.log2(max(ratio,1e-300)):6.1f}, uncertified {int(ws[0]):5d}/{N}, idx equal exact kernel: {bool(torch.equal(idx, idx_d))}")

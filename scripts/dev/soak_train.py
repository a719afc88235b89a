"""Dev tool (GPU): randomized soak of the large-batch training routes (round 3): the loss-gradient terms formed inside their consumers
(lipvq_mlp3_bwd_vq_f32, lipvq_scatter_add_sorted_vq_f32), the decoder summing the loss (lipvq_mlp3_loss_f32) and the one-launch
training forwards, against the separate launches -- every parameter gradient bit for bit, loss to 1e-7 -- over random batch sizes
from 65 536 rows, widths, codebook sizes, both tokenizers, both screens, eager and torch.use_deterministic_algorithms(True).
python scripts/dev/soak_train.py [seconds]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import lipvq_vae_amd
from lipvq_vae_amd import ops
from lipvq_vae_amd.tokenizer import LLFQVAE_V4, VQVAE
from bench import trained_like_

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(2024)
real_sup, real_loss_sup, real_sc = ops.mlp3_bwd_vq_supported, ops.mlp3_loss_supported, ops.scatter_add_vq
t0, cases, rows = time.time(), 0, 0
last_note = t0
os.environ["LIPVQ_SCREEN_MONITOR"] = "0"
while time.time() - t0 < budget:
    kind = "llfq" if rng.random() < 0.6 else "vq"
    A = int(rng.choice([3, 7, 12])); D = int(rng.choice([32, 64, 128, 208])); K = int(rng.choice([37, 128, 1000, 1024, 4096]))
    N = int(rng.choice([65536, 65537, 65568, 70001, 99999, 131072, 200003]))
    det = bool(rng.random() < 0.25)
    os.environ["LIPVQ_SCREEN_MODE"] = str(rng.choice(["fine", "coarse"]))
    torch.manual_seed(int(rng.integers(1 << 30)))
    if kind == "llfq":
        model = LLFQVAE_V4(A, D, num_codes=K).cuda(); trained_like_(model, A, seed=int(rng.integers(1 << 30)))
    else:
        model = VQVAE(A, D, num_embeddings=K).cuda()
        with torch.no_grad(): model.embedding.weight.uniform_(0.0, float(rng.choice([0.05, 0.5])))
        model.invalidate_caches()
    x = torch.randn(N, A, device="cuda") * float(rng.choice([0.3, 1.0, 3.0]))
    gs = float(rng.choice([1.0, 0.37, 5.0]))
    torch.use_deterministic_algorithms(det)
    try:
        ops.mlp3_bwd_vq_supported, ops.mlp3_loss_supported, ops.scatter_add_vq = real_sup, real_loss_sup, real_sc
        z1, l1 = model(x); (l1 * gs).backward()
        g1 = {k: v.grad.clone() for k, v in model.named_parameters()}
        model.zero_grad()
        ops.mlp3_bwd_vq_supported = lambda N, pk: False
        ops.mlp3_loss_supported = lambda N, pk: False
        ops.scatter_add_vq = lambda g, ze, table, idx, alpha, gscale=None, zq=None, deterministic=None: \
            ops.scatter_add(ops.scaled_diff(zq, ze, alpha, gscale=gscale, c=g), idx, table.shape[0], deterministic=deterministic)
        z2, l2 = model(x); (l2 * gs).backward()
    finally:
        torch.use_deterministic_algorithms(False)
        ops.mlp3_bwd_vq_supported, ops.mlp3_loss_supported, ops.scatter_add_vq = real_sup, real_loss_sup, real_sc
    tag = (kind, N, A, D, K, det, os.environ["LIPVQ_SCREEN_MODE"])
    assert torch.equal(z1, z2), ("z",) + tag
    assert abs(l1.item() - l2.item()) <= 1e-7 * abs(l2.item()), ("loss", l1.item(), l2.item()) + tag
    for k, v in model.named_parameters():
        assert torch.equal(g1[k], v.grad), (k,) + tag
    cases += 1; rows += N
    if time.time() - last_note > 60:                          # (a silent GPU run is taken to be hung after 7 minutes)
        last_note = time.time(); print(f"... {cases} cases so far", flush=True)
    del model, x, g1
print(f"soak_train: {cases} random cases, {rows} rows -- folded routes equal the separate launches bit for bit")

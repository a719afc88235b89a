"""Dev tool (GPU): randomized soak of the quantizer routes against the all-pairs exact kernel (no screen, no lists):
fused tokenize, stand-alone screened nearest, exact rows -- random shapes (round 3: any latent width 1 ... 208), seeds and
adversarial rows (duplicated codes, bisector near-ties incl. codes congruent mod 32, rows sitting on codes), both distance
rules, and per case a random screen (three-product / one-product) and fused-kernel shape.  python scripts/dev/soak.py [seconds]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import lipvq_vae_amd
from lipvq_vae_amd import ops
from lipvq_vae_amd.tokenizer import LLFQVAE_V4, VQVAE, _ScreenMonitor
from lipvq_vae_amd.ops import ACT_RELU
from bench import trained_like_

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(12345)
t0, cases, rows, unc = time.time(), 0, 0, 0
last_note = t0
while time.time() - t0 < budget:
    D = int(rng.choice([32, 64, 128, 208])) if rng.random() < 0.6 else int(rng.integers(1, 209))
    K = int(rng.choice([37, 256, 1000, 1024, 2048, 8192])) if D <= 128 else int(rng.choice([128, 1024, 4096]))
    ops.set_option("screen_mode", str(rng.choice(["fine", "coarse"])))              # read per launch by the library
    shape = rng.choice(["default", "default", "w8rg1", "w8rg2", "w4rg2", "w4rg1"])          # default: the library's size rule
    ops.set_option("tok_shape", None if shape == "default" else str(shape))
    inpl = rng.choice(["default", "0", "1"])                                                 # uncertified rows in place / listed
    ops.set_option("tok_inplace", None if inpl == "default" else str(inpl))
    ops.set_option("tok_defer_ze", str(rng.integers(2)))
    ops.set_option("tok_nt_ze", str(rng.integers(2)))
    ops.set_option("tok_ze_ring", None if rng.random() < 0.7 else "0")                       # the z_e scratch as a ring (default) / in full
    _ScreenMonitor.ENABLED = False
    A = int(rng.choice([3, 7, 12]))
    N = int(rng.choice([1, 33, 257, 2049, 4100, 30000, 70001, 100001, 300000]))
    torch.manual_seed(int(rng.integers(1 << 30)))
    model = LLFQVAE_V4(A, D, num_codes=K).cuda()
    trained_like_(model, A, seed=int(rng.integers(1 << 30)))
    cb = model.quantizer.codebook.data
    if K > 40 and rng.random() < 0.7:                       # duplicated codes (some in the same lane of the screen, some not)
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(K)); b = int((a + 32 * rng.integers(1, max(2, K // 32))) % K) if rng.random() < 0.5 else int(rng.integers(K))
            cb[b] = cb[a]
    model.invalidate_caches()
    x = torch.randn(N, A, device="cuda") * float(rng.choice([0.3, 1.0, 3.0]))
    ze = model.encode(x)
    if rng.random() < 0.6 and N > 8:                        # adversarial latents: bisectors and exact code hits, straight into the quantizers
        m = max(1, N // 4)
        ia = torch.randint(0, K, (m,), device="cuda"); ib = torch.randint(0, K, (m,), device="cuda")
        same_lane = torch.rand(m, device="cuda") < 0.5
        ib = torch.where(same_lane, (ia + 32 * torch.randint(1, max(2, K // 32), (m,), device="cuda")) % K, ib)
        t = (0.5 + (torch.randint(-3, 4, (m, 1), device="cuda").float()) * 1e-8)
        ze = ze.clone(); ze[:m] = cb[ia] * t + cb[ib] * (1 - t); ze[m:m + max(1, m // 4)] = cb[ia[:max(1, m // 4)]]
    for dist in (0, 1):                                     # the norm rule (LLFQ) and the sum rule (plain VQVAE)
        ref, _, _ = ops.nearest(ze, cb, dist=dist)
        tag = (N, A, D, K, dist, ops.get_option("screen_mode"))
        if ops.nearest_screen_supported(K, D):
            idx_s, zq_s, ws = ops.nearest_screened(ze, cb, ops.nearest_prepare(cb), return_workspace=True, dist=dist)
            assert torch.equal(idx_s, ref), ("screened",) + tag
            assert torch.equal(zq_s, cb[ref])
            unc += int(ws[0])
        if N <= (30000 if D in (32, 64, 128, 208) else 2100):          # (other widths: every row is a full scan from L2)
            idx_r, _ = ops.nearest_rows(ze, cb, dist=dist)
            assert torch.equal(idx_r, ref), ("rows",) + tag
    if ops.tokenize_supported(A, 64, 128, D, K):
        idx_f, zq_f = model.tokenize(x, count_usage=False)
        ref_f, _, _ = ops.nearest(model.encode(x), cb)
        assert torch.equal(idx_f, ref_f), ("fused", N, A, D, K, ops.get_option("screen_mode"), ops.get_option("tok_shape"), ops.get_option("tok_inplace"))
        assert torch.equal(zq_f, cb[ref_f])
    if ops.tokenize_supported(A, 64, 128, D, K) and N > 2048 and K >= 64:
        # the plain VQVAE's fused launch (ReLU instance, per-row scales) against its own encoder + the all-pairs kernel
        vq = VQVAE(A, D, num_embeddings=K).cuda()
        with torch.no_grad():
            vq.embedding.weight.copy_(cb * float(rng.choice([1.0, 0.01])))
        vq.invalidate_caches()
        idx_v, zst_v = vq.tokenize(x, count_usage=False)
        ze_v = vq.encode(x)
        ref_v, zq_v, _ = ops.nearest(ze_v, vq.embedding.weight.detach(), dist=1)
        assert torch.equal(idx_v, ref_v), ("vq fused", N, A, D, K, ops.get_option("screen_mode"))
        assert torch.equal(zst_v, ops.ste(ze_v, zq_v))
    cases += 1; rows += N
    if time.time() - last_note > 60:                          # (a silent GPU run is taken to be hung after 7 minutes)
        last_note = time.time(); print(f"... {cases} cases so far", flush=True)
print(f"soak: {cases} random cases, {rows} rows, {unc} uncertified rows through the lists -- every route equals the all-pairs exact kernel")

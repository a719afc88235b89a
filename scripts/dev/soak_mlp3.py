"""Dev soak (GPU): the large-batch (weights-in-LDS) mlp3 kernels against the small-batch kernels on the same rows sent in
pieces -- random fan-in / fan-out / activations / row counts; forward (with and without gather, saved pre-activations) and
backward-data.  Bit equality is required.  python scripts/dev/soak_mlp3.py [cases] [seed]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from lipvq_vae_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
gen = torch.Generator(device="cuda").manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), device="cuda", generator=gen).item())
ACTS = (ops.ACT_NONE, ops.ACT_GELU, ops.ACT_SIGMOID, ops.ACT_RELU)
bad = 0
for c in range(cases):
    N = 65536 + ri(0, 70000)
    K0 = [1, 3, 7, 12, 31, 32, 33, 64, 100, 128, 208, 256][ri(0, 11)]
    J2 = [1, 7, 12, 32, 33, 64, 65, 128, 208][ri(0, 8)]
    J0, J1 = ((64, 128), (128, 64))[ri(0, 1)]
    acts = tuple(ACTS[ri(0, 3)] for _ in range(3))
    W0 = torch.randn(J0, K0, device="cuda", generator=gen) * 0.4
    W1 = torch.randn(J1, J0, device="cuda", generator=gen) * 0.2
    W2 = torch.randn(J2, J1, device="cuda", generator=gen) * 0.2
    b0, b1, b2 = (torch.randn(J, device="cuda", generator=gen) for J in (J0, J1, J2))
    cut = ri(1000, 60000)
    pieces = [(0, cut)] + [(a, min(a + 60000, N)) for a in range(cut, N, 60000)]
    # forward
    pk = ops.mlp3_pack(W0, b0, W1, b1, W2, b2)
    gather = ri(0, 1) == 1
    if gather:
        table = torch.randn(500, K0, device="cuda", generator=gen)
        idx = torch.randint(0, 500, (N,), device="cuda", generator=gen)
        y, pre = ops.mlp3(table, pk, acts, gather_idx=idx, save_pre=True)
        parts = [ops.mlp3(table, pk, acts, gather_idx=idx[a:b].contiguous(), save_pre=True) for a, b in pieces]
    else:
        x = torch.randn(N, K0, device="cuda", generator=gen) * 2
        x[7] = 30.0
        y, pre = ops.mlp3(x, pk, acts, save_pre=True)
        parts = [ops.mlp3(x[a:b].contiguous(), pk, acts, save_pre=True) for a, b in pieces]
    ok = torch.equal(y, torch.cat([p[0] for p in parts])) and all(torch.equal(pre[i], torch.cat([p[1][i] for p in parts])) for i in range(3))
    # backward-data
    pkb = ops.mlp3_pack_bwd(W0, W1, W2)
    gy = torch.randn(N, J2, device="cuda", generator=gen)
    pr = [torch.randn(N, J, device="cuda", generator=gen) * 2 for J in (J0, J1, J2)]
    pr[0][11] = 12.0
    if acts[2] == ops.ACT_NONE:
        pr[2] = None
    want_gx = ri(0, 1) == 1
    big = ops.mlp3_bwd(gy, pr, pkb, acts, want_gx=want_gx)
    bp = [ops.mlp3_bwd(gy[a:b].contiguous(), [None if p is None else p[a:b].contiguous() for p in pr], pkb, acts, want_gx=want_gx)
          for a, b in pieces]
    for i in range(4):
        if big[i] is None:
            continue
        ok = ok and torch.equal(big[i], torch.cat([p[i] for p in bp]))
    if not ok:
        bad += 1
        print("FAIL", dict(N=N, K0=K0, J0=J0, J1=J1, J2=J2, acts=acts, gather=gather, want_gx=want_gx))
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)

"""Dev measurement (GPU): mlp3 backward-data at N rows, encoder and decoder chains, with the real activations and with
identity activations (what the act' VALU work costs): python scripts/dev/measure_mlp3_bwd.py [N] [D]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
A = 7
torch.manual_seed(0)
dev = "cuda"


def run(name, K0, J0, J1, J2, acts, want_gx):
    W0 = torch.randn(J0, K0, device=dev) * 0.1
    W1 = torch.randn(J1, J0, device=dev) * 0.1
    W2 = torch.randn(J2, J1, device=dev) * 0.1
    pk = ops.mlp3_pack_bwd(W0, W1, W2)
    gy = torch.randn(N, J2, device=dev)
    pre = [torch.randn(N, J0, device=dev), torch.randn(N, J1, device=dev), torch.randn(N, J2, device=dev)]
    for tag, ac in (("real acts", acts), ("identity", (ops.ACT_NONE,) * 3)):
        p = list(pre)
        if ac[2] == ops.ACT_NONE:
            p[2] = None
        for _ in range(3):
            ops.mlp3_bwd(gy, p, pk, ac, want_gx=want_gx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.mlp3_bwd(gy, p, pk, ac, want_gx=want_gx)
        e1.record()
        torch.cuda.synchronize()
        byts = 4.0 * N * (J2 * (2 if ac[2] != ops.ACT_NONE else 1) + (J2 if ac[2] != ops.ACT_NONE else 0) * 0 + 2 * J1 + 2 * J0 + (K0 if want_gx else 0)
                          + (J2 if ac[2] != ops.ACT_NONE else 0))
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:8s} {tag:10s} N={N} {K0}->{J0}->{J1}->{J2} gx={want_gx}: {ms*1e3:8.1f} us   ~{byts/ms/1e6:7.1f} GB/s algorithmic")


run("encoder", A, 64, 128, D, (ops.ACT_GELU, ops.ACT_GELU, ops.ACT_SIGMOID), False)
run("decoder", D, 64, 128, A, (ops.ACT_GELU, ops.ACT_GELU, ops.ACT_NONE), True)

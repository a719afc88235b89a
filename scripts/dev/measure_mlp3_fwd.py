"""Dev measurement (GPU): mlp3 forward (decoder stack, saving pre-activations or not) at N rows: python scripts/dev/measure_mlp3_fwd.py [N] [D]"""
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
W0, b0 = torch.randn(64, D, device=dev) * 0.1, torch.randn(64, device=dev)
W1, b1 = torch.randn(128, 64, device=dev) * 0.1, torch.randn(128, device=dev)
W2, b2 = torch.randn(A, 128, device=dev) * 0.1, torch.randn(A, device=dev)
pk = ops.mlp3_pack(W0, b0, W1, b1, W2, b2)
x = torch.randn(N, D, device=dev)
acts = (ops.ACT_GELU, ops.ACT_GELU, ops.ACT_NONE)
for save in (False, True):
    for _ in range(3):
        ops.mlp3(x, pk, acts, save_pre=save)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.mlp3(x, pk, acts, save_pre=save)
    e1.record()
    torch.cuda.synchronize()
    print(f"decoder forward N={N} D={D} save_pre={save}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")

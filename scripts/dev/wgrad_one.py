import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from lipvq_vae_amd import ops
N, J, Kd = 524288, 128, 64
G, H = torch.randn(N, J, device="cuda"), torch.randn(N, Kd, device="cuda")
for _ in range(6): ops.wgrad(G, H)
torch.cuda.synchronize()

"""Dev driver for rocprofv3 (GPU): the default action branch, eval forward (and with `train` forward + backward):
python scripts/dev/prof_default.py [N] [A] [D] [train]"""
import sys, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.default_branch import DefaultActionNetwork
warnings.simplefilter("ignore")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
A = int(sys.argv[2]) if len(sys.argv) > 2 else 12
D = int(sys.argv[3]) if len(sys.argv) > 3 else 208
train = len(sys.argv) > 4
torch.manual_seed(0)
m = DefaultActionNetwork(A, D).cuda()
x = torch.randn(N, A, device="cuda")
if train:
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for _ in range(10):
        m.zero_grad()
        m(x).square().mean().backward()
else:
    m.eval()
    with torch.no_grad():
        for _ in range(10):
            m(x)
torch.cuda.synchronize()

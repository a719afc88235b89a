import sys, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import lipvq_vae_amd
from lipvq_vae_amd.default_branch import DefaultActionNetwork
warnings.simplefilter("ignore")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
torch.manual_seed(0)
m = DefaultActionNetwork(12, 208).cuda().eval()
x = torch.randn(N, 12, device="cuda")
with torch.no_grad():
    for _ in range(10): m(x)
torch.cuda.synchronize()

"""Dev measurement (GPU box): what LLFQVAE_V4.tune finds on this device -- the four (defer_ze, nt_ze) combinations of the fused
launch at a workload's full batch and at a 65 536-row shard -- beside the round-3 tree's launch on the same device (its clock says
which kind of device this is: scripts/dev/clock_ab.py).   python scripts/dev/tune_probe.py [workload] [r03 tree]"""
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import WORKLOADS, trained_like_
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
if len(sys.argv) > 2:
    r = subprocess.run([sys.executable, str(ROOT / "scripts/dev/clock_ab.py"), wl, "1", sys.argv[2], str(ROOT)], capture_output=True, text=True)
    print("\n".join(l for l in r.stdout.splitlines() if not l.startswith("#")))
B, T, A, D, K = WORKLOADS[wl]
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
x = torch.randn(B * T, A, device="cuda")
for n in (B * T, 65536):
    for rep in range(2):
        t = model.tune(x[:n].contiguous())
        print(n, t["choice"], {k: round(v, 4) for k, v in t["ms_per_launch"].items()}, flush=True)

"""Dev tool (GPU): how the uncertified rows of a fused tokenize launch were listed (short candidate lists vs full scans)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import lipvq_vae_amd
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import WORKLOADS, trained_like_
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B, T, A, D, K = WORKLOADS[wl]
N = B * T
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
x = torch.randn(N, A, device="cuda")
model.tokenize(x); torch.cuda.synchronize()
ws = model._tok_ws.cpu().numpy()
cnt = int(ws[0]); L = (N + 15) & ~15; cap = N + 64      # lq_cand_cap (round 3: every row can have a list)
cl = ws[16 + 2 * L: 16 + 2 * L + 16 * cap].reshape(cap, 16)[:min(cnt, cap)]
n0, n1 = cl[:, 0], cl[:, 8]
full = (n0 == -1) | (n1 == -1)
lanes = ~full & ((n0 == -2) | (n1 == -2))
tot = np.where(full | lanes, -1, n0 + n1)
per_row = [bin((int(a) & 0xffff) | ((int(b) & 0xffff) << 16)).count("1") for a, b in cl[lanes][:, [1, 9]]]
print(f"{wl}: {cnt} uncertified rows; full scans {int(full.sum())}; lane scans {int(lanes.sum())}, flagged lanes per such row:",
      np.bincount(per_row) if per_row else [], "; short lists by length:", np.bincount(tot[~(full | lanes)], minlength=8)[:15])

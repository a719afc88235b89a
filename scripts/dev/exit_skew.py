"""Dev measurement (GPU, diagnostic build -DLQ_STAMPS via LIPVQ_HIP_LIBRARY): when do the workgroups of tokenize_kernel finish?
Per launch: each workgroup's end time (s_memrealtime of its last wave) relative to the first wave's start, its distribution, the
same per XCD (workgroup index mod 8), and whether the late workgroups are the same ones from launch to launch.
   python scripts/dev/exit_skew.py [workload] [rows]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import torch
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import WORKLOADS, trained_like_
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B, T, A, D, K = WORKLOADS[wl]
N = int(sys.argv[2]) if len(sys.argv) > 2 else B * T
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
x = torch.randn(N, A, device="cuda")
for _ in range(300):
    model.tokenize(x)
torch.cuda.synchronize()
NW = 4 if N <= 32768 else 8
NWG = min(256, -(-N // (NW * 32)))
off = 16 + ((N // 2) & ~1)
ends = []
for rep in range(4):
    for _ in range(20):
        model.tokenize(x)
    torch.cuda.synchronize()
    ws = model._tok_ws.cpu().numpy()
    st = ws[off:off + NWG * NW * 32].view(np.int64).reshape(NWG, NW, 16)
    t0 = st[:, :, 12].min()
    end = (st[:, :, 11].max(axis=1) - t0) / 100.0           # us, per workgroup
    start = (st[:, :, 12].min(axis=1) - t0) / 100.0
    ends.append(end)
    print(f"launch {rep}: workgroup end times us: min {end.min():.1f} p10 {np.percentile(end, 10):.1f} median {np.median(end):.1f} mean {end.mean():.1f} "
          f"p90 {np.percentile(end, 90):.1f} max {end.max():.1f}; start skew {start.max():.2f}; idle before the last one ends: mean {end.max() - end.mean():.1f} us "
          f"({100 * (end.max() - end.mean()) / end.max():.1f} % of the launch)")
    print("   per XCD (workgroup mod 8) mean end:", " ".join(f"{end[i::8].mean():.1f}" for i in range(8)))
e = np.stack(ends)
print("correlation of workgroup end times between launches:", np.round(np.corrcoef(e)[0, 1:], 2))
late = [set(np.argsort(v)[-16:]) for v in e]
print("of the 16 latest workgroups of launch 0, also among the 16 latest of launches 1..3:", [len(late[0] & s) for s in late[1:]])

"""Dev measurement (GPU, diagnostic build -DLQ_STAMPS): where a wave of tokenize_kernel spends its cycles.
   python scripts/stamps.py [workload]     (library from LIPVQ_HIP_LIBRARY if set)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import lipvq_vae_amd
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import WORKLOADS, trained_like_
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B, T, A, D, K = WORKLOADS[wl]
N = int(sys.argv[2]) if len(sys.argv) > 2 else B * T           # optional: a shard of that many rows
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
x = torch.randn(N, A, device="cuda")
for _ in range(20):
    model.tokenize(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); model.tokenize(x); e1.record(); torch.cuda.synchronize()
ws = model._tok_ws.cpu().numpy()
off = 16 + ((N // 2) & ~1)
import os
_shape = lipvq_vae_amd._capi.get_option("tok_shape") or ("w4rg1" if N <= 32768 else "w8rg1")   # (the library's size rule, lipvq_fused.hip::tok_shape)
NW = 4 if _shape.startswith("w4") else 8      # waves per workgroup of the instance that ran
# (options come from the environment only with LIPVQ_DEV_KNOBS=1: lipvq-vae_amd/_capi.py)
NWG = min(256, -(-N // (NW * 32)))                              # workgroups of the launch
st = ws[off:off + NWG * NW * 32].view(np.int64).reshape(NWG * NW, 16)
names = ["layer0+gelu", "layer1", "layer2+finish", "scale/split", "screen loop", "decide", "gather+idx", "-"]
tot = st[:, 8].astype(np.float64)
print(f"{wl}: launch {e0.elapsed_time(e1):.3f} ms; wave lifetime median {np.median(tot):.0f} ticks (s_memtime), min {tot.min():.0f} max {tot.max():.0f}")
rt = (st[:, 11].max() - st[:, 12].min()) / 100.0                # s_memrealtime: 100 MHz, common to all CUs
clk = np.median((st[:, 8] + st[:, 10]) / np.maximum(1, st[:, 11] - st[:, 12])) * 100.0
print(f" prologue (kernel entry -> first row block) median {np.median(st[:, 10]):.0f} ticks; kernel, first wave in -> last wave out "
      f"{rt:.1f} us; entry skew over waves {(st[:, 12].max() - st[:, 12].min()) / 100.0:.2f} us; exit skew {(st[:, 11].max() - st[:, 11].min()) / 100.0:.2f} us; "
      f"shader clock {clk:.0f} MHz")
for half, sel in ((("waves 0-3", np.arange(NWG * 8) % 8 < 4), ("waves 4-7", np.arange(NWG * 8) % 8 >= 4)) if NW == 8 else
                  (("waves 0-3", np.arange(NWG * 4) >= 0),)):
    m = st[sel]
    print(f" {half}: lifetime {np.median(m[:, 8]):.0f}; start skew vs wave 0 of the launch {np.median(m[:, 9] - st[:, 9].min()):.0f}")
    for i, n in enumerate(names[:7]):
        print(f"    {n:14s} {np.median(m[:, i]):10.0f}  ({100 * np.median(m[:, i]) / np.median(m[:, 8]):5.1f} %)")
    print(f"    {'unaccounted':14s} {np.median(m[:, 8] - m[:, :7].sum(1)):10.0f}")
    print(f"    {'(sync in loop)':14s} {np.median(m[:, 7]):10.0f}")

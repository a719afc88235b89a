#!/usr/bin/env python3
"""Time the sibling bin tokenizer (csrc/lipvq_bin.hip) on the GPU: per-kernel HIP-event timings and the torch-eager
sequence the reference runs (bucketize per dimension, A embedding lookups, cat, Linear, GELU, Linear, GELU).

    python scripts/measure_bin.py [--rows 524288] [--A 7] [--D 64] [--iters 20]
"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lipvq_vae_amd import ops  # noqa: E402
from lipvq_vae_amd.binning import AdaptiveBinActionEmbedding  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=524288)
    ap.add_argument("--A", type=int, default=7)
    ap.add_argument("--D", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    N, A, D = a.rows, a.A, a.D
    m = AdaptiveBinActionEmbedding(A, D).cuda().eval()
    x = torch.randn((N, A), device="cuda")
    nb, H = m.num_bins, 32 * A
    with torch.no_grad():
        m(x)
        m._update_enabled = False
        t_all = timed(lambda: m(x), a.iters)
        tmin, tmax = m.running_min.clone(), m.running_max.clone()
        t_mm = timed(lambda: ops.bin_minmax(x, tmin, tmax), a.iters)
        bins = ops.bin_discretize(x, m.running_min, m.running_max, nb)
        t_disc = timed(lambda: ops.bin_discretize(x, m.running_min, m.running_max, nb), a.iters)
        l1, l2 = m.output_layer[0], m.output_layer[2]
        P = torch.stack([ops.linear(e.weight, l1.weight[:, 64 * i:64 * (i + 1)].contiguous()) for i, e in enumerate(m.embedding_layers)])
        h = ops.bin_hidden(bins, P, l1.bias)
        t_hid = timed(lambda: ops.bin_hidden(bins, P, l1.bias), a.iters)
        t_lin = timed(lambda: ops.linear(h, l2.weight, l2.bias, act=ops.ACT_GELU), a.iters)

        def torch_path():
            idx = []
            for i in range(A):
                bd = torch.linspace(m.running_min[i], m.running_max[i], nb + 1, device="cuda")
                idx.append(torch.clamp(torch.bucketize(x[:, i].contiguous(), bd) - 1, 0, nb - 1))
            emb = torch.cat([m.embedding_layers[i](idx[i]) for i in range(A)], dim=-1)
            return m.output_layer(emb)
        t_torch = timed(torch_path, max(3, a.iters // 4))
    print(json.dumps({"what": "AdaptiveBinActionEmbedding.forward (stats frozen)", "rows": N, "A": A, "D": D, "us": t_all * 1e6,
                      "actions_per_s": N / t_all}))
    print(json.dumps({"kernel": "bin_minmax_kernel", "us": t_mm * 1e6, "GBps": N * A * 4 / t_mm / 1e9}))
    print(json.dumps({"kernel": "bin_discretize_kernel", "us": t_disc * 1e6, "GBps": N * A * 12 / t_disc / 1e9}))
    print(json.dumps({"kernel": "bin_hidden_kernel", "us": t_hid * 1e6, "GBps_written": N * H * 4 / t_hid / 1e9,
                      "lds_gathers_per_s": N * H * A / t_hid}))
    print(json.dumps({"kernel": "linear_kernel+gelu", "us": t_lin * 1e6, "TFLOPs": 2.0 * N * H * D / t_lin / 1e12}))
    print(json.dumps({"what": "torch eager sequence of the reference (same GPU)", "us": t_torch * 1e6, "speedup": t_torch / t_all}))


if __name__ == "__main__":
    main()

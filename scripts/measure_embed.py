#!/usr/bin/env python3
"""Time the embedding stage after the tokenizer (csrc/lipvq_embed.hip) on the GPU and price it against its roofline.

embed_rows_kernel is HBM-write bound: algorithmic bytes per action = 4 E (the output row) + 8 (its index); the [K][E]
table and the [T][E] time embeddings are read from L2.  Prints one JSON line per configuration.

    python scripts/measure_embed.py [--rows 524288] [--E 512] [--T 10] [--K 1024] [--iters 50]
"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lipvq_vae_amd import ops  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(fn, iters):
    for _ in range(5):
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
    ap.add_argument("--E", type=int, default=512)
    ap.add_argument("--T", type=int, default=10)
    ap.add_argument("--K", type=int, default=1024)
    ap.add_argument("--D", type=int, default=64)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--contig", action="store_true", help="write [N][E] contiguously instead of into the 2t+1 slots")
    a = ap.parse_args()
    N = a.rows // a.T * a.T
    dev = "cuda"
    cb = torch.rand((a.K, a.D), device=dev)
    W = torch.randn((a.E, a.D), device=dev) / a.D ** 0.5
    b = torch.randn(a.E, device=dev)
    pos = 0.02 * torch.randn((a.T, a.E), device=dev)
    lw, lb = torch.ones(a.E, device=dev), torch.zeros(a.E, device=dev)
    idx = torch.randint(0, a.K, (N,), device=dev)
    out = torch.empty((N // a.T, 3 * a.T, a.E), device=dev)
    table = ops.linear(cb, W, b)
    t_table = timed(lambda: ops.linear(cb, W, b), a.iters)
    if a.contig:
        t_rows = timed(lambda: ops.embed_rows(table, idx, pos, lw, lb, 1e-5, out, N, a.T, a.T * a.E, a.E, 0), a.iters)
    else:
        t_rows = timed(lambda: ops.embed_rows(table, idx, pos, lw, lb, 1e-5, out, N, a.T, 3 * a.T * a.E, 2 * a.E, a.E), a.iters)
    bytes_rows = N * (4 * a.E + 8)
    print(json.dumps({"kernel": "embed_rows_kernel", "contig": a.contig, "rows": N, "E": a.E, "T": a.T, "K": a.K, "us": t_rows * 1e6,
                      "actions_per_s": N / t_rows, "algorithmic_GBps": bytes_rows / t_rows / 1e9,
                      "frac_of_hbm_peak": bytes_rows / t_rows / 1e9 / HBM_PEAK_GBS}))
    print(json.dumps({"kernel": "linear_kernel(table)", "K": a.K, "D": a.D, "E": a.E, "us": t_table * 1e6,
                      "TFLOPs": 2.0 * a.K * a.D * a.E / t_table / 1e12}))
    flat = out.view(-1)[:N * a.E]
    t_fill = timed(lambda: flat.fill_(1.0), a.iters)
    print(json.dumps({"kernel": "torch fill_ of the same bytes (store-bandwidth calibration)", "us": t_fill * 1e6,
                      "GBps": N * a.E * 4 / t_fill / 1e9}))
    # what the reference does instead: gather z_q rows, cuBLAS-style Linear over N rows, add, LayerNorm, stack/view/cat
    zq = cb[idx].view(N // a.T, a.T, a.D)
    ln = torch.nn.LayerNorm(a.E).to(dev)

    def torch_path():
        e = torch.nn.functional.linear(zq, W, b) + pos
        return ln(e)
    t_torch = timed(torch_path, max(5, a.iters // 5))
    print(json.dumps({"kernel": "torch eager (Linear+add+LayerNorm, no interleave)", "us": t_torch * 1e6,
                      "speedup_of_embed_rows": t_torch / t_rows}))
    # dense stream (observation rows): Linear over N rows + embed_rows with idx = NULL
    Nd = min(N, 65536) // a.T * a.T
    xd = torch.randn((Nd, a.D), device=dev)
    t_lin = timed(lambda: ops.linear(xd, W, b), a.iters)
    print(json.dumps({"kernel": "linear_kernel(dense)", "rows": Nd, "Kin": a.D, "E": a.E, "us": t_lin * 1e6,
                      "TFLOPs": 2.0 * Nd * a.D * a.E / t_lin / 1e12}))


    for (n, k, e) in ((524288, 224, 64), (131072, 384, 208), (65536, 512, 512)):
        xx, ww, bb2 = torch.randn((n, k), device=dev), torch.randn((e, k), device=dev), torch.randn(e, device=dev)
        t = timed(lambda: ops.linear(xx, ww, bb2, act=ops.ACT_GELU), max(5, a.iters // 5))
        print(json.dumps({"kernel": "linear+gelu", "rows": n, "Kin": k, "E": e, "us": t * 1e6, "TFLOPs": 2.0 * n * k * e / t / 1e12}))


if __name__ == "__main__":
    main()

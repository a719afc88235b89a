#!/usr/bin/env python3
"""bench.py -- actions tokenized / s (encode + quantize) on 1..8 MI355X.

A "step" is one pass of the tokenizer hot path (LLFQVAE_V4.tokenize: encoder MLP -> Lipschitz
latent layer -> nearest code -> z_latent gather + code-usage histogram) over one synthetic
batch that is already resident in HBM.  Workload (BASELINE.json configs[1]): B=4096, T=128,
A=7, codebook K=1024 x D=64, fp32 everywhere (the parity mode: indices are bit-identical to
the CPU oracle).  With N GPUs every rank tokenizes its own B x T batch (weak scaling, rows are
independent) and the per-step code-usage histogram [K] int64 is all-reduced over RCCL -- the
path's only cross-GPU dependency.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the fused tokenize launch: algorithmic flops per launch (164 736 flop
                per row at config 2) / mean duration from HIP events on its stream, against the
                blended matrix-pipe peak of its instruction mix (encoder on the fp32 MFMA pipe,
                distance screen on the fp16 MFMA pipe with 3 split products per algorithmic
                product); `frac_of_fp32_peak` relates the same flops to the 157.3 TFLOP/s fp32
                peak that SURVEY.md 8d prices the exact all-pairs scan against.
  cpu_baseline  the torch-CPU restatement of the reference (oracle/lipvq_oracle.py,
                kind="port"), timed on a bounded row sample on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

WORKLOADS = {
    # name: (B, T, A, D, K)
    "cfg2": (4096, 128, 7, 64, 1024),
    "cfg3": (4096, 128, 7, 128, 8192),
    "cfg1": (64, 16, 7, 32, 256),
    "encA": (4096, 128, 7, 64, 32),      # dev only: tiny codebook -> the fused kernel is almost pure encoder phase
    "icrt": (4096, 128, 12, 208, 1024),  # the reference's own widths (obs_nets.py:2411, v5:89-92) at the metric's batch: unfused path
}
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 vector = fp32-input MFMA peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (the screening kernel's pipe)
PEAK_HBM_GBS = 8000.0


def trained_like_(model, A, seed=0):
    """Put a freshly constructed LLFQVAE_V4 into the trained-like regime of SURVEY.md 8d, using
    the product path itself for the z_e samples (no oracle import on the measured path)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    K, D = model.num_codes, model.latent_dim
    dev = model.quantizer.codebook.device
    with torch.no_grad():
        model.to_latent.ci.fill_(40.0)
        model.to_latent.b.copy_(torch.randn(D, generator=g).to(dev))
        cb = torch.rand(K, D, generator=g).to(dev)
        xs = torch.randn(max(4 * K, 1024), A, generator=g).to(dev)
        ze = model.encode(xs)
        pick = torch.randperm(xs.shape[0], generator=g)[: K // 2].to(dev)
        cb[: K // 2] = ze[pick] + 0.02 * torch.randn(K // 2, D, generator=g).to(dev)
        model.quantizer.codebook.copy_(cb)


def cpu_baseline(model, x_dev, idx_dev, budget_s=12.0):
    """Time the torch-CPU restatement on a bounded sample of the same batch; check index parity."""
    from oracle import lipvq_oracle as O
    p = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    x = x_dev.cpu()
    # a 1-GPU box shares its host: use the CPUs this process may run on, capped at the box's
    # per-GPU share (16); torch's default (all 256 hardware threads) oversubscribes and is slower
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))
    torch.set_num_threads(threads)
    chunk = 256 if model.num_codes * model.latent_dim <= 1024 * 64 else 32
    # calibrate on a small slice, then size the sample for ~budget_s of CPU work
    n0 = 4 * chunk
    t = time.perf_counter()
    O.torch_llfq_tokenize(p, x[:n0], chunk=chunk)
    rate0 = n0 / (time.perf_counter() - t)
    n = int(min(x.shape[0], max(n0, (rate0 * budget_s) // chunk * chunk)))
    t = time.perf_counter()
    idx_cpu, _ = O.torch_llfq_tokenize(p, x[:n], chunk=chunk)
    dt = time.perf_counter() - t
    idx_gpu = idx_dev[:n].cpu()
    bad = torch.nonzero(idx_cpu != idx_gpu).reshape(-1)
    mism = int(bad.numel())
    # A mismatch can only be a near-tie: the GPU equals the canonical oracle bit for bit, whose encoder differs from
    # torch's MKL/Sleef arithmetic by ~3e-7 in z_e.  Quantify: relative gap between the two candidates' distances,
    # evaluated with the reference's own (torch-CPU) z_e.
    worst_gap = 0.0
    if mism:
        with torch.no_grad():
            ze = O.torch_llfq_encode(p, x[bad])
            cb = p["quantizer.codebook"]
            da = torch.norm(ze - cb[idx_cpu[bad]], dim=-1)
            db = torch.norm(ze - cb[idx_gpu[bad]], dim=-1)
            worst_gap = float(((db - da).abs() / torch.maximum(da, db)).max())
    # what the reference's train() actually sets (scripts/train.py:57): one thread, on a ~3 s slice
    torch.set_num_threads(1)
    n1 = int(max(chunk, min(n, (rate0 / max(1, threads) * 3.0) // chunk * chunk)))
    t = time.perf_counter()
    O.torch_llfq_tokenize(p, x[:n1], chunk=chunk)
    rate_1t = n1 / (time.perf_counter() - t)
    torch.set_num_threads(threads)
    return {
        "value": n / dt, "unit": "actions/s", "cores": threads, "kind": "port", "value_1_thread": rate_1t,
        "sample": f"first {n} rows of the same batch, torch-CPU restatement, {chunk}-row chunks, {dt:.1f} s",
        "host_cpus": os.cpu_count(), "index_mismatches_vs_gpu": mism, "rows_compared": n,
        "max_rel_distance_gap_of_mismatches": worst_gap,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if os.environ.get("LIPVQ_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # "nccl" is RCCL on ROCm.  LIPVQ_BENCH_BACKEND=gloo exists only to rehearse the multi-process
        # flow on a box with fewer GPUs than ranks (ranks then share cuda:0).
        backend = os.environ.get("LIPVQ_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import lipvq_vae_amd  # noqa: F401  (fails loudly if the HIP library is missing)
    from lipvq_vae_amd import ops
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4

    B, T, A, D, K = WORKLOADS[args.workload]
    N = B * T
    torch.manual_seed(0)
    model = LLFQVAE_V4(A, D, num_codes=K).to(dev)
    trained_like_(model, A, seed=0)
    gx = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = torch.randn(B, T, A, generator=gx).to(dev).reshape(N, A)     # flattened as tensor_utils.py:1066-1067 does

    # two histograms: the all-reduce of step k runs asynchronously (RCCL's own stream) under step k+1
    ubuf = [torch.zeros_like(model.code_usage), torch.zeros_like(model.code_usage)]
    pending = [None, None]
    ev_pairs = []
    step_no = [0]

    def step(timed):
        b = step_no[0] & 1
        step_no[0] += 1
        if pending[b] is not None:
            pending[b].wait()               # the stream waits for that reduction before the buffer is reused
            pending[b] = None
        ubuf[b].zero_()
        model.code_usage = ubuf[b]
        # == LLFQVAE_V4.tokenize: ONE fused launch (encoder + Lipschitz layer + MFMA screen, csrc/lipvq_fused.hip)
        # followed by the exact kernel for the rows the screen could not certify.  HIP events bracket it
        # on the stream it is launched on (torch's current stream is handed to the C ABI).
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        idx, zq = model.tokenize(x)
        if timed:
            e1.record()
            ev_pairs.append((e0, e1))
        if dist is not None:
            pending[b] = dist.all_reduce(ubuf[b], async_op=True)   # global code-usage histogram (8 KiB for K=1024)
        return idx, zq

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for _ in range(args.warmup):
        step(False)
    drain()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        idx, zq = step(True)
    drain()                                 # every histogram reduction is inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    tok_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) / max(1, len(ev_pairs))
    value = world * N * args.steps / elapsed
    # algorithmic work per launch: encoder 2*(A*64+64*128+128*D) + distance 2*K*D flop per row
    enc_flop = 2.0 * N * (A * 64 + 64 * 128 + 128 * D)
    dist_flop = 2.0 * N * K * D
    algo_flop = enc_flop + dist_flop
    # the kernel executes the encoder on the fp32 matrix pipe (157.3 TF/s) and the distance screen on the
    # fp16 matrix pipe with 3 split products per algorithmic product (2500 TF/s dense): its floor is the sum
    # of both pipe times, and the peak it is priced against is the algorithmic flops over that floor
    t_floor = enc_flop / (PEAK_FP32_TFLOPS * 1e12) + 3.0 * dist_flop / (PEAK_F16_MFMA_TFLOPS * 1e12)
    peak_blend = algo_flop / t_floor / 1e12
    achieved = algo_flop / (tok_ms * 1e-3) / 1e12 if tok_ms > 0 else 0.0
    exact_rows = int(model.last_exact_rows[0]) if model.last_exact_rows is not None else None
    traffic = None
    tfile = ROOT / "profiles" / "hbm_traffic.json"           # PMC-derived bytes per launch (separate rocprofv3 --pmc runs)
    if tfile.exists():
        t = json.loads(tfile.read_text())
        if t.get("workload") == args.workload:
            traffic = t.get("bytes_per_launch")
    out = {
        "metric": "actions tokenized/sec (encode+quantize) at B=4096 T=128 K=1024, 1/2/4/8 GPU",
        "value": value, "unit": "actions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: B={B} T={T} action_dim={A} codebook K={K} d={D}, "
                               f"fp32 encoder + fp32 argmin (parity mode: indices bit-identical to the CPU oracle), "
                               f"per-GPU batch fixed",
                   "rows_per_gpu": N, "parallelism": f"batch-sharded x{world}, all-reduce of code usage [K] int64"},
        "roofline": {"bound": "mfma", "kernel": "tokenize_kernel (+ nearest_rows_kernel for uncertified rows)",
                     "achieved": achieved, "peak": peak_blend, "unit": "TFLOP/s", "frac": achieved / peak_blend,
                     "traffic": traffic, "ms_per_launch": tok_ms, "algorithmic_flop_per_launch": algo_flop,
                     "floor_ms": t_floor * 1e3,
                     "peak_note": "algorithmic flop / (encoder flop / 157.3 TF/s fp32 MFMA + 3 x distance flop / 2500 TF/s fp16 MFMA)",
                     "frac_of_fp32_peak": achieved / PEAK_FP32_TFLOPS,
                     "algorithmic_bytes_per_launch": float(N) * (4 * A + 4 * D + 8),
                     "rows_decided_by_exact_kernel": exact_rows},
    }
    if world == 1 and ops.tokenize_supported(A, 64, model.hidden_dim, D, K):
        # reported beside the metric, never as `value`: the opt-in fast mode (fp16 encoder GEMMs, fp32 accumulation and
        # quantizer) on the same batch, with the fraction of indices that differ from the parity run above
        idx_parity = idx.clone()
        for _ in range(3):
            model.tokenize(x, count_usage=False, mode="fast")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            idx_fast, _ = model.tokenize(x, count_usage=False, mode="fast")
        e1.record()
        torch.cuda.synchronize()
        fast_ms = e0.elapsed_time(e1) / args.steps
        out["fast_mode"] = {"value": N / (fast_ms * 1e-3), "unit": "actions/s", "ms_per_step": fast_ms,
                            "dtype": "f16 encoder operands, f32 accumulation, f32 quantizer",
                            "index_flip_rate_vs_parity": float((idx_fast != idx_parity).float().mean().item()),
                            "note": "opt-in (tokenize(mode='fast')); not bit-identical, hence not the reported value"}
    if world == 1:
        # "report also full fwd (+decode+loss) and fwd+bwd+AdamW step" -- same batch, a few steps each,
        # beside the metric (never `value`)
        def timed_ms(fn, n):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for _ in range(n):
                fn()
            a1.record()
            torch.cuda.synchronize()
            return a0.elapsed_time(a1) / n

        def full_forward():
            with torch.no_grad():
                model(x)

        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)       # icl.py:885-889

        def train_step():
            opt.zero_grad()
            _, loss = model(x)
            loss.backward()
            opt.step()

        saved = {k: v.clone() for k, v in model.state_dict().items()}
        ff_ms = timed_ms(full_forward, 5)
        ts_ms = timed_ms(train_step, 5)
        model.load_state_dict(saved)                 # the cpu_baseline below compares against the untrained parameters
        out["also"] = {"full_forward": {"value": N / (ff_ms * 1e-3), "unit": "actions/s", "ms_per_step": ff_ms,
                                        "what": "forward(x) without autograd: encode + quantize + decode + three losses"},
                       "train_step": {"value": N / (ts_ms * 1e-3), "unit": "actions/s", "ms_per_step": ts_ms,
                                      "what": "zero_grad + forward + loss.backward() + AdamW.step() (icl.py:913-914, 968-970)"}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, x, idx)
        out["gpu_vs_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

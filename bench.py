#!/usr/bin/env python3
"""bench.py -- actions tokenized / s (encode + quantize) on 1..8 MI355X.

A "step" is one pass of the tokenizer hot path (LLFQVAE_V4.tokenize: encoder MLP -> Lipschitz
latent layer -> nearest code -> z_latent gather + code-usage histogram) over one synthetic
batch that is already resident in HBM.  Workload (BASELINE.json configs[1]): B=4096, T=128,
A=7, codebook K=1024 x D=64, fp32 everywhere (the parity mode: indices are bit-identical to
the CPU oracle).  With N GPUs (BASELINE config 4) the GLOBAL batch stays B x T and is sharded
B/N sequences per rank (`--scaling strong`, the default: value = global rows / max-over-ranks time); `--scaling weak`
keeps the per-GPU batch at B x T instead.  Rows are independent: no data-path collective; the per-step code-usage
histogram [K] int64 is all-reduced over RCCL -- the path's only cross-GPU dependency.

Launching: `python bench.py --gpus N` from a plain shell starts the N ranks itself (the parent makes no
GPU call; it runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process, relays
rank 0's line and exits with the children's status).  Under an external launcher (RANK / WORLD_SIZE in the
environment, as the driver's `python -m torch.distributed.run ... bench.py --gpus N` sets them) it runs as one rank.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the fused tokenize launch: algorithmic flops per launch (164 736 flop
                per row at config 2) / mean launch duration (one pair of HIP events over the K timed steps, on the
                launches' stream: first launch's start to last launch's end, / K), against two floors:
                `frac` = the matrix-pipe floor of its instruction mix (encoder on the fp32 MFMA pipe,
                distance screen on the fp16 MFMA pipe with 3 split products per algorithmic product);
                `frac_algorithmic_floor` = the stricter reading, algorithmic 2NKD distance flops at the
                fp16 MFMA peak with no credit for the 3x split.
  parity_gate   rank 0, N=1: the timed batch's indices against the torch-CPU restatement of the reference on
                the cpu_baseline sample.  A mismatch whose two candidates' distances differ by a relative gap
                >= 1e-6 FAILS the run: no `value`, non-zero exit.
  sustained     >= 1000 back-to-back launches (>= 0.5 s of GPU time) beside the K-step reading, so the
                number is not a burst-clock figure.
  cpu_baseline  the torch-CPU restatement of the reference (oracle/lipvq_oracle.py, kind="port"), timed on
                a bounded row sample on this box's host cores at the best thread count of a short sweep.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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


PARITY_GAP = 1e-6     # a CPU/GPU index mismatch is tolerated only between candidates closer than this (relative distance gap)


def cpu_baseline(model, x_dev, idx_dev, budget_s=10.0):
    """Time the torch-CPU restatement on a bounded sample of the same batch (best thread count of a short sweep)
    and compare its indices with the GPU's on that sample."""
    from oracle import lipvq_oracle as O
    p = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    x = x_dev.cpu()
    chunk = 256 if model.num_codes * model.latent_dim <= 1024 * 64 else 32
    # a 1-GPU box shares its host: sweep thread counts up to the CPUs this process may run on, capped at the box's
    # per-GPU share (16).  One thread is what the reference's train() sets (scripts/train.py:57) and is often the
    # fastest on 256-row chunks (more threads oversubscribe the small [chunk, K, D] temporaries).
    avail = max(1, min(len(os.sched_getaffinity(0)), 16))
    sweep = {}
    n0 = 4 * chunk
    for th in sorted({1, min(4, avail), min(8, avail), avail}):
        torch.set_num_threads(th)
        O.torch_llfq_tokenize(p, x[:chunk], chunk=chunk)               # touch
        t = time.perf_counter()
        n_s = 0
        while time.perf_counter() - t < 1.0:
            O.torch_llfq_tokenize(p, x[n_s:n_s + n0], chunk=chunk)
            n_s += n0
        sweep[th] = n_s / (time.perf_counter() - t)
    best_th = max(sweep, key=sweep.get)
    torch.set_num_threads(best_th)
    n = int(min(x.shape[0], max(n0, (sweep[best_th] * budget_s) // chunk * chunk)))
    t = time.perf_counter()
    idx_cpu, _ = O.torch_llfq_tokenize(p, x[:n], chunk=chunk)
    dt = time.perf_counter() - t
    idx_gpu = idx_dev[:n].cpu()
    bad = torch.nonzero(idx_cpu != idx_gpu).reshape(-1)
    mism = int(bad.numel())
    # A mismatch can only be a near-tie: the GPU equals the canonical oracle bit for bit, whose encoder differs from
    # torch's MKL/Sleef arithmetic by ~5e-7 in z_e.  Quantify: relative gap between the two candidates' distances,
    # evaluated with the reference's own (torch-CPU) z_e.
    worst_gap = 0.0
    if mism:
        with torch.no_grad():
            ze = O.torch_llfq_encode(p, x[bad])
            cb = p["quantizer.codebook"]
            da = torch.norm(ze - cb[idx_cpu[bad]], dim=-1)
            db = torch.norm(ze - cb[idx_gpu[bad]], dim=-1)
            worst_gap = float(((db - da).abs() / torch.maximum(da, db)).max())
    base = {
        "value": n / dt, "unit": "actions/s", "cores": best_th, "kind": "port",
        "thread_sweep_actions_per_s": {str(k): v for k, v in sweep.items()},
        "sample": f"first {n} rows of the same batch, torch-CPU restatement of the reference, {chunk}-row chunks, "
                  f"{best_th} thread(s) (best of the sweep), {dt:.1f} s",
        "host_cpus": os.cpu_count(), "cpus_available": avail,
    }
    gate = {"rows_compared": n, "index_mismatches": mism, "max_rel_distance_gap_of_mismatches": worst_gap,
            "tolerated_gap": PARITY_GAP, "passed": bool(mism == 0 or worst_gap < PARITY_GAP),
            "against": "torch-CPU restatement of the reference (bit-identical to the reference module in-process)"}
    return base, gate


CHILD_SCHEDULE = []          # the counter children run the schedule THIS process tuned (main sets it): ["--schedule", "d,n"]


def pmc_child_runs(workload, passes, child_steps=5, sustained=200, timeout_s=150):
    """Counter passes of the metric's kernels, measured NOW: one child run of this file under `rocprofv3 --pmc <counters>` per pass
    (counters only -- no trace domain -- and the program directly behind `--`), after the timed region so that the profiler never
    touches `value`.  The child warms the chip the way the headline run does (a sustained loop, then warm-up), and only the LAST
    `child_steps` dispatches of each kernel -- its timed steps -- are read.  Returns {pass name: {kernel: {counter: mean value,
    "_ns": mean dispatch duration, "_n": dispatches read}}}."""
    import collections
    import csv
    import glob
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        raise RuntimeError("rocprofv3 not found")
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        raise RuntimeError("this process is itself running under a profiler; no nested counter run")
    tmp = tempfile.mkdtemp(prefix="lipvq_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    child = [sys.executable, str(ROOT / "bench.py"), "--workload", workload, "--steps", str(child_steps), "--warmup", "2",
             "--no-cpu-baseline", "--sustained", str(sustained), "--metric-only", "--traffic", "off"] + CHILD_SCHEDULE
    result = {}
    try:
        for name, counters in passes:
            d = os.path.join(tmp, name)
            # the profiler and the bench child it starts run in a process group of their own: on a timeout the whole
            # group is killed and reaped, so no orphan keeps the GPU busy after this function returns
            pr = subprocess.Popen([rp, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                out_b, _ = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                pr.communicate()
                raise RuntimeError(f"rocprofv3 --pmc {' '.join(counters)} timed out after {timeout_s} s (process group killed)")
            if pr.returncode != 0:
                raise RuntimeError(f"rocprofv3 --pmc {' '.join(counters)} exited {pr.returncode}: {out_b.decode(errors='replace')[-300:]}")
            acc = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> [(start, value, ns)]
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] in counters:
                            t0_, t1_ = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
                            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append((t0_, float(row["Counter_Value"]), t1_ - t0_))
            res = {}
            for k, byc in acc.items():
                res[k] = {}
                for cn, v in byc.items():
                    v.sort()
                    last = v[-child_steps:]
                    res[k][cn] = sum(x[1] for x in last) / len(last)
                    res[k]["_ns"] = sum(x[2] for x in last) / len(last)
                    res[k]["_n"] = len(last)
            result[name] = res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return result


def measure_traffic_live(workload, kernels, timeout_s=150):
    """HBM bytes per launch of the metric's kernels (FETCH_SIZE and WRITE_SIZE in separate passes) and, in a third pass, the SQ /
    GRBM counters the roofline is read against: matrix-pipe busy cycles and the shader clock the launch actually held.  Unit and
    gfx950 correction as MI355X_MICROARCH.md's HBM section prescribes: both size counters are in KiB; FETCH_SIZE under-reports wide
    reads by 2x on gfx950.  Returns (bytes_per_launch, source, detail, sq) or raises."""
    runs = pmc_child_runs(workload, (("fetch", ("FETCH_SIZE",)), ("write", ("WRITE_SIZE",)),
                                     ("sq", ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))),
                          timeout_s=timeout_s)
    detail, total = {}, 0.0
    for k in sorted(set(runs["fetch"]) | set(runs["write"])):
        if not any(t in k for t in kernels):
            continue
        f, w = runs["fetch"].get(k, {}), runs["write"].get(k, {})
        rb, wb = 2.0 * f.get("FETCH_SIZE", 0.0) * 1024.0, w.get("WRITE_SIZE", 0.0) * 1024.0
        detail[k] = {"dispatches": max(f.get("_n", 0), w.get("_n", 0)), "read_bytes": rb, "write_bytes": wb}
        total += rb + wb
    if not detail:
        raise RuntimeError("no dispatch of the metric's kernels in the counter output")
    sq = None
    dom = [k for k in runs["sq"] if kernels[0] in k]
    if dom:
        c = runs["sq"][max(dom, key=lambda k: runs["sq"][k].get("_ns", 0.0))]
        gui, ns = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("_ns", 0.0)
        if gui > 0 and ns > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1 024 SIMDs (MI355X_MICROARCH.md, DVFS give-back)
            sq = {"kernel": max(dom, key=lambda k: runs["sq"][k].get("_ns", 0.0)), "dispatches": c.get("_n"),
                  "ns_per_dispatch_profiled": ns, "shader_clock_mhz": gui / 8.0 / ns * 1e3,
                  "mfma_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * 8.0 / (1024.0 * gui),
                  "insts_mfma": c.get("SQ_INSTS_MFMA"), "sq_busy_cycles": c.get("SQ_BUSY_CYCLES"), "grbm_gui_active": gui,
                  "how": "third --pmc pass of the same child run (sustained loop first; the last 5 dispatches = its timed steps): clock = "
                         "GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration, mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
                         "GRBM_GUI_ACTIVE / 8)"}
    return total, ("live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child runs of bench.py --sustained 200 --steps 5 "
                   "--metric-only on this GPU, after the timed region); mean over the child's timed steps, summed over the launch's "
                   "kernels; reads x2 (gfx950 FETCH_SIZE correction)"), detail, sq


def sample_power(run_some, seconds=2.5):
    """Socket power / shader clock / junction temperature as rocm-smi reports them WHILE the metric's launch loops (after the timed
    region; rank 0, one GPU).  The launch runs at or near the package power limit on most boxes of the pool and the clock is what
    that leaves (profiles/r04_i_clock_ab.txt): the reading says which kind of box a line came from.  Returns a dict or None."""
    import threading
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True,
                                   timeout=10)
                card = json.loads(r.stdout)
                card = card[sorted(card)[0]]
                rec = {}
                for k, v in card.items():
                    kl = k.lower()
                    if kl.startswith("sclk clock speed"):
                        rec["sclk"] = float(str(v).strip("()Mhz "))
                    elif "power" in kl and "(w)" in kl:
                        rec["power"] = float(v)
                    elif "temperature" in kl and "junction" in kl:
                        rec["temp"] = float(v)
                samples.append(rec)
            except Exception:  # noqa: BLE001 -- optional equipment
                pass
            stop.wait(0.15)
    th = threading.Thread(target=sampler, daemon=True)
    t_end = time.time() + seconds
    th.start()
    while time.time() < t_end:
        run_some()
    stop.set()
    th.join(timeout=15)
    samples = samples[1:] if len(samples) > 2 else samples           # (the first sample may predate the loop)

    def mean(key):
        v = [s_[key] for s_ in samples if key in s_]
        return sum(v) / len(v) if v else None
    if not samples or mean("power") is None:
        return None
    return {"socket_power_w": mean("power"), "sclk_mhz": mean("sclk"), "junction_temp_c": mean("temp"), "samples": len(samples),
            "how": f"rocm-smi sampled every ~0.3 s while the metric's launch looped for {seconds} s after the timed region"}


def self_launch(args):
    """Plain `python bench.py --gpus N`: start N fresh ranks.  This process has made no GPU call (importing torch does
    not initialise HIP) and makes none; the ranks are children of torch.distributed.run, never an exec of this process."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    # stdout is inherited: rank 0's JSON line goes straight through; torchrun's own chatter is on stderr
    rc = subprocess.call(cmd, env=env)
    raise SystemExit(rc)


def rehearse(args, rank, world, backend):
    """Launcher rehearsal (tests/test_bench_launcher.py, CPU, gloo): the whole multi-rank flow of a bench run -- rendezvous,
    double-buffered asynchronous all-reduce of a [K] int64 histogram under the next step, barrier-bracketed timing, MAX over
    ranks, one line from rank 0 -- with a host-side stand-in for the tokenizer launch.  It measures nothing: the line carries
    "rehearsal": true and no metric/value, so it can never be read as a bench result."""
    import torch.distributed as dist
    dist.init_process_group(backend)
    from lipvq_vae_amd.sharded import shard_bounds       # host logic only (the rehearsal makes no GPU call)
    B, K = WORKLOADS[args.workload][0], WORKLOADS[args.workload][4]
    # the sharding of the real run: strong = this rank's share of the workload's B sequences, weak = B sequences per rank
    b_lo, b_hi = shard_bounds(B, rank, world) if args.scaling == "strong" else (0, B)
    seqs = torch.zeros(1, dtype=torch.int64)
    ubuf = [torch.zeros(K, dtype=torch.int64), torch.zeros(K, dtype=torch.int64)]
    pending = [None, None]
    total = torch.zeros(K, dtype=torch.int64)
    dist.barrier()
    t0 = time.perf_counter()
    for s in range(args.warmup + args.steps):
        b = s & 1
        if pending[b] is not None:
            pending[b].wait()
            total += ubuf[b]
        ubuf[b].zero_()
        ubuf[b][(rank + s) % K] += 1 + rank           # stand-in for the kernel's histogram of this rank's rows
        seqs += b_hi - b_lo
        pending[b] = dist.all_reduce(ubuf[b], async_op=True)
    for b in (0, 1):
        if pending[b] is not None:
            pending[b].wait()
            total += ubuf[b]
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(seqs)
    if rank == 0:
        print(json.dumps({"rehearsal": True, "scaling": args.scaling, "sequences_per_step_all_ranks": int(seqs.item()) // (args.warmup + args.steps),
                          "n_gpus": world, "world_size": dist.get_world_size(), "backend": dist.get_backend(),
                          "steps": args.steps, "warmup": args.warmup, "usage_sum": int(total.sum()),
                          "expected_usage_sum": (args.warmup + args.steps) * world * (world + 1) // 2,
                          "elapsed_max_s": float(tmax.item())}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"),
                    help="N > 1: strong = the workload's global batch sharded B/N per rank (BASELINE config 4 as SURVEY 8d defines "
                         "it); weak = every rank tokenizes a full B x T batch of its own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustained", type=int, default=1000, help="back-to-back launches of the sustained reading (0 = skip)")
    ap.add_argument("--traffic", default="live", choices=("live", "file", "off"),
                    help="roofline.traffic: measured now under rocprofv3 --pmc (N=1 only), read from profiles/hbm_traffic.json, or null")
    ap.add_argument("--schedule", default=None, help="defer_ze,nt_ze (e.g. 0,1): fix the fused launch's schedule choices instead of tuning")
    ap.add_argument("--no-tune", action="store_true",
                    help="skip LLFQVAE_V4.tune (set-up, untimed): the fused launch's device-dependent schedule choices stay at their defaults")
    ap.add_argument("--metric-only", action="store_true",
                    help="skip the fast-mode / full-forward / training-step side readings (profiling runs: every dispatch is then the metric's)")
    ap.add_argument("--rehearse-launcher", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and args.gpus > 1:
        self_launch(args)                   # does not return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    backend = os.environ.get("LIPVQ_BENCH_BACKEND", "nccl")
    if args.rehearse_launcher:
        return rehearse(args, rank, world, backend)
    # stdout carries ONE JSON line.  RCCL (and gloo) print banners to the C stdout when a communicator is created ("RCCL version :
    # ...", "[Gloo] Rank 0 is connected ..."): with more than one rank, file descriptor 1 points at stderr until the line is printed.
    saved_stdout = None
    if world > 1 or os.environ.get("LIPVQ_BENCH_FORCE_DIST") == "1":
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # LIPVQ_BENCH_FORCE_DIST=1: initialise the process group even with ONE rank (a one-GPU box can then exercise the whole
    # collective path of the N > 1 run -- RCCL communicator, asynchronous bucket all-reduce, barriers -- with a world of one)
    if world > 1 or os.environ.get("LIPVQ_BENCH_FORCE_DIST") == "1":
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        # "nccl" is RCCL on ROCm.  LIPVQ_BENCH_BACKEND=gloo exists only to rehearse the multi-process
        # flow on a box with fewer GPUs than ranks (ranks then share cuda:0).
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import lipvq_vae_amd  # noqa: F401  (fails loudly if the HIP library is missing)
    from lipvq_vae_amd import ops
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4

    from lipvq_vae_amd.sharded import shard_batch, shard_bounds

    B, T, A, D, K = WORKLOADS[args.workload]
    strong = args.scaling == "strong"
    if strong and B < world:
        raise SystemExit(f"--scaling strong: {B} sequences cannot be split over {world} ranks")
    torch.manual_seed(0)
    model = LLFQVAE_V4(A, D, num_codes=K).to(dev)
    trained_like_(model, A, seed=0)
    if strong:
        # every rank draws the SAME global [B, T, A] batch and keeps its own B/world sequences (sharded.shard_bounds):
        # the union over the ranks is the 1-GPU workload, row for row
        xg = torch.randn(B, T, A, generator=torch.Generator(device="cpu").manual_seed(1234))
        x = shard_batch(xg, rank, world).to(dev).contiguous()        # [B_local * T, A], flattened as tensor_utils.py:1066-1067 does
        del xg
        N_global = B * T
    else:
        gx = torch.Generator(device="cpu").manual_seed(1234 + rank)
        x = torch.randn(B, T, A, generator=gx).to(dev).reshape(B * T, A)
        N_global = world * B * T
    N = x.shape[0]                                                   # rows THIS rank tokenizes per step
    b_lo, b_hi = shard_bounds(B, rank, world) if strong else (0, B)

    # Per-step code-usage histograms, all-reduced in BUCKETS of M steps: two sets of M rows [M][K] int64; every step counts into
    # its own row, after M steps ONE asynchronous all-reduce covers the set (32 KiB for M = 4, K = 1024) while the next M steps
    # fill the other set.  Every step's GLOBAL histogram still exists (M steps later at most); what changes is how often a
    # collective kernel has to find a CU on a chip whose every CU holds a persistent tokenizer workgroup (two waves of 256
    # registers per SIMD leave no room beside it, so a collective launched per step either delays that step's launch or waits
    # for its end: ~25-30 us of a 440 us step).  xGMI rings are per-link bound: fewer, larger messages.
    # LIPVQ_BENCH_COLLECTIVE=capi routes it through the library's own RCCL binding (lipvq_allreduce_counts, include/lipvq.h)
    # instead of torch.distributed's process group (the default: the same RCCL underneath).
    # Strong scaling shortens the step with the shard (65 536 rows per rank at N = 8), so the bucket grows with the world size:
    # the collective's share of the wall clock stays what it is at one GPU's step length.
    M = max(1, int(os.environ.get("LIPVQ_BENCH_USAGE_BUCKET", str(min(16, 4 * world) if strong else 4))))
    ubuf = [torch.zeros((M,) + tuple(model.code_usage.shape), dtype=model.code_usage.dtype, device=dev) for _ in range(2)]
    pending = [None, None]
    capi_comm = None
    if dist is not None and os.environ.get("LIPVQ_BENCH_COLLECTIVE", "torch") == "capi" and backend == "nccl":
        from lipvq_vae_amd.sharded import RcclCounts
        capi_comm = RcclCounts()

    def wait_pending(b):
        if pending[b] is not None:
            if capi_comm is not None:
                capi_comm.wait(pending[b])
            else:
                pending[b].wait()
            pending[b] = None
    ev_pairs = []
    step_no = [0]
    last_row = [None]

    def reduce_set(b):
        if capi_comm is not None:
            pending[b] = capi_comm.all_reduce(ubuf[b].view(-1))
        elif dist is not None:
            pending[b] = dist.all_reduce(ubuf[b], async_op=True)    # global code-usage histograms of M steps

    def step(timed):
        s_ = step_no[0]
        b, m = (s_ // M) & 1, s_ % M
        step_no[0] += 1
        if m == 0:
            wait_pending(b)                 # the stream waits for that set's reduction before its rows are reused
            ubuf[b].zero_()                 # ONE fill per set of M steps (a fill per step is a 2-3 us launch: 4 % of a 65 536-row shard's step)
        row = ubuf[b][m]
        model.code_usage = row
        last_row[0] = row
        # == LLFQVAE_V4.tokenize: ONE fused launch (encoder + Lipschitz layer + MFMA screen, csrc/lipvq_fused.hip)
        # followed by the exact kernel for the rows the screen could not certify (none where the launch decides them in place).
        idx, zq = model.tokenize(x)
        if m == M - 1:
            reduce_set(b)
        return idx, zq

    def flush_partial():
        """A set that is only partly filled when a timed region ends is reduced as well (rows not yet written are zeros or old
        rows: harmless, they are rewritten before they are read)."""
        s_ = step_no[0]
        if s_ % M != 0:
            b = (s_ // M) & 1
            reduce_set(b)
            step_no[0] = (s_ // M + 1) * M         # the next step starts a fresh set

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def drain():
        flush_partial()
        for b in (0, 1):
            wait_pending(b)

    # Set-up, untimed: every rank lets the library measure the fused launch's device-dependent schedule choices on ITS device with
    # ITS shard (lipvq_tokenize_tune_f32, ~0.3 s: identical results under every choice; MI355X devices hold different clocks under
    # the same kernel and what wins on one loses on another, profiles/r04_i_clock_ab.txt).  The line reports what was chosen.
    tuned = None
    if args.schedule:
        d_, n_ = (v.strip() for v in args.schedule.split(","))
        ops.set_option("tok_defer_ze", d_)
        ops.set_option("tok_nt_ze", n_)
        tuned = {"choice": {"defer_ze": int(d_), "nt_ze": int(n_)}, "fixed": "--schedule"}
    elif not args.no_tune:
        tuned = model.tune(x)
        fence()
    if tuned is not None:
        CHILD_SCHEDULE[:] = ["--schedule", f"{tuned['choice']['defer_ze']},{tuned['choice']['nt_ze']}"]

    # Sustained reading FIRST: the same step, >= 0.4 s of back-to-back launches (same barriers, MAX over ranks).  It is a
    # number of its own (`sustained`), and it leaves the chip in the clock / power state of a job that has been running for a
    # while -- which is what the W warm-up + K timed steps below are then measured in.  (Round 1 timed K = 20 steps straight
    # after model set-up: 11 ms on a GPU that had been idle, 12-15 % below the steady rate of the very same launches.)
    sustained = None
    if args.sustained > 0:
        for _ in range(3):
            step(False)
        drain()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.sustained):
            step(False)
        drain()
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        sustained = {"launches": args.sustained, "seconds": el, "ms_per_step": 1e3 * el / args.sustained,
                     "value": N_global * args.sustained / el, "unit": "actions/s",
                     "order": "measured before the warm-up + timed steps of `value` (it also serves as their clock warm-up)"}

    for _ in range(args.warmup):
        step(False)
    drain()
    fence()
    # ONE pair of HIP events over the timed region, on the stream the launches go to (torch's current stream is handed to the C ABI):
    # mean launch duration = their distance / K.  (Until round 4 every step carried its own pair: forty event packets in a
    # twenty-step region, 12 us of queue gaps per step that the sustained loop does not have -- the instrument slowed the measurement.)
    e_first, e_last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e_first.record()
    for _ in range(args.steps):
        idx, zq = step(True)
    e_last.record()
    ev_pairs.append((e_first, e_last))
    drain()                                 # every histogram reduction is inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # every rank's last histogram is the GLOBAL one: N_global rows
        usage_rows = int(last_row[0].sum().item())
    else:
        usage_rows = int(last_row[0].sum().item())
    idx_timed = idx.clone()

    tok_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) / max(1, args.steps)      # (first launch's start -> last launch's end) / K
    value = N_global * args.steps / elapsed

    # algorithmic work per launch: encoder 2*(A*64+64*128+128*D) + distance 2*K*D flop per row
    enc_flop = 2.0 * N * (A * 64 + 64 * 128 + 128 * D)
    dist_flop = 2.0 * N * K * D
    algo_flop = enc_flop + dist_flop
    # floor 1 (instruction mix): the kernel executes the encoder on the fp32 matrix pipe (157.3 TF/s) and the distance
    # screen on the fp16 matrix pipe with 3 split products per algorithmic product (2500 TF/s dense)
    # (the one-product screen -- wide latents against large codebooks, e.g. cfg3 -- executes ONE product: no credit then)
    coarse = bool(ops.screen_is_coarse(K, D)) and ops.tokenize_supported(A, 64, model.hidden_dim, D, K)
    split = 1.0 if coarse else 3.0
    t_floor = enc_flop / (PEAK_FP32_TFLOPS * 1e12) + split * dist_flop / (PEAK_F16_MFMA_TFLOPS * 1e12)
    # floor 2 (strict): algorithmic distance flops at the fp16 pipe, no credit for the 3x split
    t_floor_alg = enc_flop / (PEAK_FP32_TFLOPS * 1e12) + dist_flop / (PEAK_F16_MFMA_TFLOPS * 1e12)
    peak_blend = algo_flop / t_floor / 1e12
    achieved = algo_flop / (tok_ms * 1e-3) / 1e12 if tok_ms > 0 else 0.0
    exact_rows = int(model.last_exact_rows[0]) if model.last_exact_rows is not None else None
    fused = ops.tokenize_supported(A, 64, model.hidden_dim, D, K)
    be = (f"{dist.get_backend()} world_size={dist.get_world_size()}, collective via "
          f"{'lipvq_allreduce_counts (C ABI -> RCCL)' if capi_comm is not None else 'torch.distributed'}"
          if dist is not None else "single process")
    out = {
        "metric": "actions tokenized/sec (encode+quantize) at B=4096 T=128 K=1024, 1/2/4/8 GPU",
        "value": value, "unit": "actions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: B={B} T={T} action_dim={A} codebook K={K} d={D}, "
                               f"fp32 encoder + fp32 argmin (parity mode; distance screen on fp16 MFMA, exact fp32 re-scoring), "
                               + (f"global batch fixed: {B} sequences sharded {b_hi - b_lo} per GPU" if strong else
                                  f"per-GPU batch fixed at {B} sequences (global {world * B})"),
                   "rows_per_gpu": N, "global_rows": N_global,
                   "parallelism": f"batch-sharded x{world} ({be}), per-step code usage [K] int64 all-reduced in "
                                  f"buckets of {M} steps, overlapped with the following steps",
                   "global_usage_rows_last_step": usage_rows},
        "roofline": {"bound": "mfma",
                     "kernel": (("tokenize_kernel, one-product screen (+ nearest_lists_kernel / nearest_rows_kernel: the exact stage over "
                                 "the rows it leaves)" if coarse else
                                 "tokenize_kernel (+ nearest_lists_kernel / nearest_rows_kernel: the exact stage over the uncertified rows, read back from "
                                 "the z_e the launch stored)") if fused else
                                "mlp3_wg_kernel + screen_kernel (+ nearest_rows_kernel for uncertified rows)"),
                     "screen": "one fp16 product per algorithmic product (11-bit operands, lower-bound bookkeeping)" if coarse else
                               "three fp16 products per algorithmic product (22-bit operands)",
                     "achieved": achieved, "peak": peak_blend, "unit": "TFLOP/s", "frac": achieved / peak_blend,
                     "frac_algorithmic_floor": (t_floor_alg * 1e3) / tok_ms if tok_ms > 0 else 0.0,
                     "traffic": None, "traffic_source": None, "traffic_detail": None,
                     "ms_per_launch": tok_ms, "algorithmic_flop_per_launch": algo_flop,
                     "floor_ms": t_floor * 1e3, "floor_algorithmic_ms": t_floor_alg * 1e3,
                     "peak_note": f"frac: algorithmic flop / (encoder flop / 157.3 TF/s fp32 MFMA + {split:g} x distance flop / 2500 TF/s "
                                  "fp16 MFMA: the products the screen executes); frac_algorithmic_floor: the same with 1 x (no credit "
                                  "for a split)",
                     "algorithmic_bytes_per_launch": float(N) * (4 * A + 4 * D + 8),
                     # SURVEY 8d asks for both fractions: the same launch against the HBM roof (it is compute bound by a wide margin)
                     "hbm_frac": (float(N) * (4 * A + 4 * D + 8) / (tok_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if tok_ms > 0 else 0.0,
                     "rows_decided_by_exact_kernel": exact_rows},
    }
    if sustained is not None:
        out["sustained"] = sustained
    out["tuned"] = tuned if tuned is not None else "not tuned (--no-tune, no fused launch for this shape, or a latent width above 64 whose instances have no device-dependent choice): library defaults"
    if tuned is not None and "fixed" not in tuned:
        tuned["what"] = ("set-up, untimed, rank 0's device: LLFQVAE_V4.tune timed the fused launch's four (defer_ze, nt_ze) schedule "
                         "combinations (identical results) and the process keeps the fastest")
    if world == 1 and not args.metric_only and ops.tokenize_fast_supported(A, 64, model.hidden_dim, D, K):
        # reported beside the metric, never as `value`: the opt-in fast mode (fp16 encoder GEMMs, fp32 accumulation and
        # quantizer) on the same batch, with the fraction of indices that differ from the parity run above
        # Same discipline as the metric: a clock warm-up of the kernel being timed, then a loop long enough not to be a burst
        # reading (round 2 timed 20 launches right behind three warm-ups: 0.381 ms, while 200-launch loops of the same build on
        # the same box read 0.330 -- profiles/r03_a_fast_mode_bisect.txt).
        nf = max(args.steps, 200)
        for _ in range(100):
            model.tokenize(x, count_usage=False, mode="fast")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nf):
            idx_fast, _ = model.tokenize(x, count_usage=False, mode="fast")
        e1.record()
        torch.cuda.synchronize()
        fast_ms = e0.elapsed_time(e1) / nf
        out["fast_mode"] = {"value": N / (fast_ms * 1e-3), "unit": "actions/s", "ms_per_step": fast_ms,
                            "dtype": "f16 encoder operands, f32 accumulation, f32 quantizer",
                            "index_flip_rate_vs_parity": float((idx_fast != idx_timed).float().mean().item()),
                            "launches_timed": nf,
                            "note": "opt-in (tokenize(mode='fast')); not bit-identical, hence not the reported value"}
    if world == 1 and not args.metric_only:
        # "report also full fwd (+decode+loss) and fwd+bwd+AdamW step" -- same batch, a few steps each,
        # beside the metric (never `value`)
        def timed_ms(fn, n):
            for _ in range(4):              # (the first steps size the allocator's pools: a 2-step warm-up once read 3.2 ms for 2.8)
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

        # icl.py:885-889's optimizer as the library's trainer builds it (icl.VQTokenizerTrainer): torch.optim.AdamW's state and
        # arithmetic, step() in two launches for the whole parameter list (optim.py) instead of torch's eight to ten
        from lipvq_vae_amd.optim import AdamW
        opt = AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)

        def train_step():
            opt.zero_grad()
            _, loss = model(x)
            loss.backward()
            opt.step()

        saved = {k: v.clone() for k, v in model.state_dict().items()}
        ff_ms = timed_ms(full_forward, 10)
        ts_ms = timed_ms(train_step, 10)
        model.load_state_dict(saved)                 # the cpu_baseline below compares against the untrained parameters
        # denominators for the two side readings (the same pricing as `roofline`: fp32 work at the fp32 MFMA peak, the distance
        # screen's three split products at the fp16 MFMA peak; HBM is not the binding roof for either)
        dec_flop = 2.0 * N * (D * 64 + 64 * 128 + 128 * A)
        ff_floor = (enc_flop + dec_flop) / (PEAK_FP32_TFLOPS * 1e12) + split * dist_flop / (PEAK_F16_MFMA_TFLOPS * 1e12)
        # training: forward + backward-data (the encoder's input gradient is not needed: its first layer drops out) + weight
        # gradients = 3x the two stacks' forward flops minus that one layer
        ts_floor = (3.0 * (enc_flop + dec_flop) - 2.0 * N * A * 64) / (PEAK_FP32_TFLOPS * 1e12) + 3.0 * dist_flop / (PEAK_F16_MFMA_TFLOPS * 1e12)
        out["also"] = {"full_forward": {"value": N / (ff_ms * 1e-3), "unit": "actions/s", "ms_per_step": ff_ms,
                                        "floor_ms": ff_floor * 1e3, "frac": ff_floor * 1e3 / ff_ms,
                                        "what": "forward(x) without autograd: encode + quantize + decode + three losses"},
                       "train_step": {"value": N / (ts_ms * 1e-3), "unit": "actions/s", "ms_per_step": ts_ms,
                                      "floor_ms": ts_floor * 1e3, "frac": ts_floor * 1e3 / ts_ms,
                                      "what": "zero_grad + forward + loss.backward() + AdamW.step() (icl.py:913-914, 968-970); floor = "
                                              "forward + backward-data + weight-gradient flops of both stacks at 157.3 TF/s fp32 MFMA "
                                              "+ the screen's 3 split products at 2500 TF/s"}}
    # HBM traffic LAST among the GPU readings: the profiler children run after every timed side reading (and are reaped before
    # this process goes on), so nothing of them can overlap a measurement
    traffic, traffic_src, traffic_detail, sq = None, None, None, None
    if args.traffic == "live" and world == 1 and not args.metric_only:
        try:
            traffic, traffic_src, traffic_detail, sq = measure_traffic_live(
                args.workload, ("tokenize_kernel", "nearest_rows", "nearest_lists") if ops.tokenize_supported(A, 64, model.hidden_dim, D, K)
                else ("mlp3_wg_kernel", "mlp3_lds_kernel", "screen_kernel", "nearest_rows", "nearest_lists"))
        except Exception as e:  # noqa: BLE001 -- the profiler is optional equipment; the committed measurement stands in
            traffic_src = f"live measurement failed ({type(e).__name__}: {str(e)[:200]}); "
    tfile = ROOT / "profiles" / "hbm_traffic.json"           # PMC-derived bytes per launch (separate rocprofv3 --pmc runs)
    if traffic is None and args.traffic != "off" and tfile.exists():
        t = json.loads(tfile.read_text())
        if t.get("workload") == args.workload:
            traffic, traffic_src = t.get("bytes_per_launch"), (traffic_src or "") + "committed: " + str(t.get("source"))
    if world == 1 and rank == 0 and not args.metric_only and args.traffic != "off":
        def _some():
            for _ in range(200):
                step(False)
            drain()
            fence()
        pw = sample_power(_some)
        if pw is not None:
            out["roofline"]["power"] = pw
    if sq is not None:
        # the same fraction against peaks scaled to the clock the launch held (the datasheet peaks are 2.4 GHz figures)
        out["roofline"].update({"mfma_busy_frac": sq["mfma_busy_frac"], "shader_clock_mhz": sq["shader_clock_mhz"],
                                "frac_at_measured_clock": out["roofline"]["frac"] * 2400.0 / sq["shader_clock_mhz"],
                                "frac_algorithmic_floor_at_measured_clock": out["roofline"]["frac_algorithmic_floor"] * 2400.0 / sq["shader_clock_mhz"],
                                "sq_counters": sq})
    out["roofline"].update({"traffic": traffic, "traffic_source": traffic_src, "traffic_detail": traffic_detail,
                            "traffic_note": "includes N x D x 4 bytes of z_e STORES beyond the algorithmic bytes (134 MB at cfg2): the launch keeps "
                                            "z_e for its exact stage -- deciding the uncertified rows from stored rows beats re-encoding them from "
                                            "x (profiles/r03_z_ze_store_ab.txt); where the rows are decided in place the stores go to a 16 MB ring, "
                                            "but the L2 still writes them through (profiles/r04_n_ze_ring_ab.txt); per kernel and per cause: "
                                            "profiles/r04_e_traffic_accounting.md "
                                            "(z_e / z_q rows are nontemporal stores, for which WRITE_SIZE reads 5-15 % above the bytes stored)"})
    failed = False
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], gate = cpu_baseline(model, x, idx_timed)
        out["parity_gate"] = gate
        out["gpu_vs_cpu"] = value / out["cpu_baseline"]["value"]
        out["config"]["workload"] += (f"; indices bit-identical to the reference's CPU arithmetic except near-ties below "
                                      f"{PARITY_GAP:g} relative distance gap (n = {gate['index_mismatches']} of "
                                      f"{gate['rows_compared']} rows compared)")
        if not gate["passed"]:
            # a real disagreement with the reference: the throughput number means nothing -- withhold it and fail
            failed = True
            out["value"] = None
            out["error"] = (f"parity gate failed: {gate['index_mismatches']} index mismatches vs the reference CPU path, "
                            f"largest relative distance gap {gate['max_rel_distance_gap_of_mismatches']:.3g} >= {PARITY_GAP:g}")
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        sys.stdout.flush()
        os.dup2(2, 1)                       # (whatever the teardown prints is not part of the result)
        dist.destroy_process_group()
    if failed:
        raise SystemExit(3)


if __name__ == "__main__":
    main()

"""Dev measurement (ONE GPU): does a collective's kernel find a CU next to the persistent tokenizer grid?  (VERDICT r3 #4b)

bench.py's step, 1 000 launches back to back, with a side-stream job per bucket of M steps issued while the next tokenize launch
is in flight -- exactly the bench's pattern -- and the tokenizer grid at 256 workgroups (every CU taken) or 252 (four left free):
   side = none      the tokenizer alone
   side = rccl      lipvq_allreduce_counts on a world-1 communicator (sharded.RcclCounts: side stream + event)
   side = torch     torch.distributed all_reduce(async_op=True) on a world-1 nccl group
   side = kernel    a stand-in that surely launches a kernel on a side stream: an in-place add over the [M][K] int64 bucket (32 KiB)
A world-1 RCCL all-reduce may be a no-op; `kernel` is the conservative reading of what a real collective kernel would face.
   python scripts/dev/rccl_contention.py [workload] [launches]      (the grid is set through lipvq_set_option("tok_grid"))"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
import lipvq_vae_amd  # noqa: F401
from lipvq_vae_amd import _capi
from lipvq_vae_amd.sharded import RcclCounts
from lipvq_vae_amd.tokenizer import LLFQVAE_V4
from bench import WORKLOADS, trained_like_

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rows = int(sys.argv[3]) if len(sys.argv) > 3 else None
B, T, A, D, K = WORKLOADS[wl]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(0)
model = LLFQVAE_V4(A, D, num_codes=K).cuda()
trained_like_(model, A)
x = torch.randn(B * T, A, generator=torch.Generator(device="cpu").manual_seed(1234)).cuda()
if rows:
    x = x[:rows].contiguous()
M = 4
ubuf = [torch.zeros(M, K, dtype=torch.int64, device=dev) for _ in range(2)]
comm = RcclCounts()
side_stream = torch.cuda.Stream()


def run(side, grid):
    _capi.set_option("tok_grid", str(grid) if grid else None)
    pending = [None, None]

    def wait(b):
        p = pending[b]
        if p is None:
            return
        if side == "rccl":
            comm.wait(p)
        elif side == "torch":
            p.wait()
        elif side == "kernel":
            torch.cuda.current_stream().wait_event(p)
        pending[b] = None

    def reduce(b):
        if side == "rccl":
            pending[b] = comm.all_reduce(ubuf[b].view(-1))
        elif side == "torch":
            pending[b] = dist.all_reduce(ubuf[b], async_op=True)
        elif side == "kernel":
            side_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side_stream):
                ubuf[b].add_(1)
                ev = torch.cuda.Event()
                ev.record()
            pending[b] = ev

    def loop(n):
        for s in range(n):
            b, m = (s // M) & 1, s % M
            if m == 0:
                wait(b)
            row = ubuf[b][m]
            row.zero_()
            model.code_usage = row
            model.tokenize(x)
            if m == M - 1:
                reduce(b)
        for b in (0, 1):
            wait(b)

    loop(100)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        loop(L)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / L)
    return best


print(f"{wl}: {x.shape[0]} rows per launch, {L} launches per reading (best of 3), usage bucket of {M} steps; ms per step")
print(f"{'side job':>10} {'grid 256':>10} {'grid 252':>10} {'grid 248':>10}")
for side in ("none", "rccl", "torch", "kernel"):
    r = [run(side, g) for g in (0, 252, 248)]
    print(f"{side:>10} {r[0]:10.4f} {r[1]:10.4f} {r[2]:10.4f}", flush=True)
dist.destroy_process_group()

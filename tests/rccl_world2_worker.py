"""Worker of tests/test_gpu_comm.py::test_world2_rccl (not a test module): one rank of a 2-rank job.
   python tests/rccl_world2_worker.py <rank> <world> <port> <backend: nccl|gloo>
Each rank tokenizes its shard of one global batch on ITS GPU (backend nccl: cuda:<rank>; the gloo rehearsal shares cuda:0),
then the per-batch code-usage histogram crosses the ranks twice -- through torch.distributed (RCCL when the backend is nccl)
and through the library's own C-ABI collective (lipvq_allreduce_counts, RcclCounts; nccl only) -- and both must equal the
histogram a single process gets from the whole batch; the shards' indices concatenate to the single-process indices."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, backend = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    devno = rank if backend == "nccl" else 0
    torch.cuda.set_device(devno)
    dev = torch.device("cuda", devno)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    import lipvq_vae_amd  # noqa: F401
    from lipvq_vae_amd.sharded import RcclCounts, ShardedTokenizer, shard_bounds
    from lipvq_vae_amd.tokenizer import LLFQVAE_V4
    from oracle import lipvq_oracle as O

    A, D, K, B, T = 7, 64, 1024, 96, 50                       # 4 800 rows: the fused launch + a few uncertified rows
    p = O.make_params(91, A, D, K, oracle=O.CanonicalOracle())
    model = LLFQVAE_V4(A, D, num_codes=K).to(dev)
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.items()})
    xg = torch.from_numpy(O.make_inputs(92, B * T, A)).reshape(B, T, A).to(dev)

    # single-process answer (every rank computes it: same parameters, same batch)
    model.code_usage.zero_()
    idx_full, _ = model.tokenize(xg.reshape(B * T, A))
    usage_full = model.code_usage.clone()
    assert int(usage_full.sum()) == B * T

    # (a) ShardedTokenizer: shard + delta all-reduce through torch.distributed
    model.code_usage.zero_()
    st = ShardedTokenizer(model)
    idx_loc, z_loc = st.tokenize(xg)
    assert torch.equal(model.code_usage, usage_full), "torch.distributed all-reduce: global histogram differs"
    s, e = shard_bounds(B, rank, world)
    assert torch.equal(idx_loc.reshape(-1), idx_full[s * T:e * T]), "a shard's indices differ from the full batch's"
    parts = [torch.empty((shard_bounds(B, r, world)[1] - shard_bounds(B, r, world)[0]) * T, dtype=torch.int64, device=dev)
             for r in range(world)]
    if backend == "nccl":
        dist.all_gather(parts, idx_loc.reshape(-1).contiguous())
    else:
        cpu_parts = [t.cpu() for t in parts]
        dist.all_gather(cpu_parts, idx_loc.reshape(-1).cpu().contiguous())
        parts = [t.to(dev) for t in cpu_parts]
    assert torch.equal(torch.cat(parts), idx_full), "the shards do not concatenate to the single-process result"

    # (b) the library's own collective (C ABI -> RCCL): only with one GPU per rank
    if backend == "nccl":
        rc = RcclCounts()
        assert rc.world == world and rc.rank == rank
        model.code_usage.zero_()
        model.tokenize(xg[s:e].reshape(-1, A))
        local = model.code_usage.clone()
        for rep in range(3):                                   # back to back, compute queued behind the wait
            buf = local.clone()
            rc.wait(rc.all_reduce(buf))
            torch.cuda.synchronize()
            assert torch.equal(buf, usage_full), f"lipvq_allreduce_counts: global histogram differs (call {rep})"
        g = torch.full((4099,), float(rank + 1), device=dev)
        rc.wait(rc.all_reduce_f32(g))
        torch.cuda.synchronize()
        assert torch.equal(g, torch.full_like(g, world * (world + 1) / 2))
        rc.close()
    # (c) bench.py's pattern (round 4, VERDICT r3 #4a): per-step histograms in two sets of M rows [M][K]; after M steps ONE
    # asynchronous all-reduce covers the set while the next M tokenize launches fill the other one -- through torch.distributed
    # and (nccl only) through lipvq_allreduce_counts on its side stream.  Every step's row must come back as the GLOBAL histogram.
    M, steps = 4, 24
    xs = xg[s:e].reshape(-1, A).contiguous()
    routes = ["torch"] + (["capi"] if backend == "nccl" else [])
    for route in routes:
        rc2 = RcclCounts() if route == "capi" else None
        ubuf = [torch.zeros(M, K, dtype=torch.int64, device=dev) for _ in range(2)]
        pending, checked = [None, None], 0

        def wait_set(b):
            nonlocal checked
            if pending[b] is None:
                return
            if rc2 is not None:
                rc2.wait(pending[b])
            else:
                pending[b].wait()
            pending[b] = None
            torch.cuda.current_stream().synchronize()
            for m in range(M):
                assert torch.equal(ubuf[b][m], usage_full), f"{route}: bucket row {m} is not the global histogram"
                checked += 1

        for st_ in range(steps):
            b, m = (st_ // M) & 1, st_ % M
            if m == 0:
                wait_set(b)                                    # that set's reduction is done before its rows are reused
            row = ubuf[b][m]
            row.zero_()
            model.code_usage = row
            model.tokenize(xs)                                 # launch k+1 in flight ...
            if m == M - 1:                                     # ... while the set just filled is reduced
                pending[b] = rc2.all_reduce(ubuf[b].view(-1)) if rc2 is not None else dist.all_reduce(ubuf[b], async_op=True)
        for b in (0, 1):
            wait_set(b)
        assert checked == steps, (route, checked)
        if rc2 is not None:
            rc2.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} {backend}: ok", flush=True)


if __name__ == "__main__":
    main()

"""Multi-GPU use of the tokenizer: one process per GPU, rows sharded, parameters replicated.

The path is row-independent given the parameters (every op of reference backbone_lfqvae_v5.py:70-76 is
row-wise), so a batch [B, T, A] is split over the ranks along B with NO data-path collective.  What does
cross ranks (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests):

  * ``all_reduce_usage``      the per-code usage histogram [K] int64 -- 8 KiB at K=1024: latency bound,
                              one small all-reduce per batch;
  * ``all_reduce_gradients``  data-parallel training: the reference's losses are means over the GLOBAL
                              row count (v5:79-81), so each rank's gradients are weighted by
                              n_local / n_global and summed in ONE flat fp32 buffer (14 tensors,
                              ~375 KiB at config 2) before AdamW.step(); replicas stay identical.

  * ``all_reduce_ema_stats``  opt-in EMA extension: int64 counts [K] + fp32 per-code sums [K, D].

Nothing here computes on the data: it only shards, flattens and calls torch.distributed.
"""
from __future__ import annotations

from typing import Iterable, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of rank's contiguous share of n items; the first n % world ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_batch(actions: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """This rank's sequences of a [B, T, A] batch, flattened to [B_local * T, A] the way the reference
    flattens before the tokenizer (robomimic/utils/tensor_utils.py:1066-1067)."""
    if actions.dim() != 3:
        raise ValueError(f"expected [B, T, A], got {tuple(actions.shape)}")
    s, e = shard_bounds(actions.shape[0], rank, world)
    return actions[s:e].reshape(-1, actions.shape[2])


def all_reduce_usage(usage: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the code-usage histogram over the ranks, in place."""
    if usage.dtype != torch.int64:
        raise TypeError("usage histogram must be int64")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if usage.is_cuda and dist.get_backend(group) == "gloo":
            # rehearsal transport (ranks sharing one GPU): gloo moves host memory; 8 KiB through the host and back
            host = usage.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            usage.copy_(host)
        else:
            dist.all_reduce(usage, op=dist.ReduceOp.SUM, group=group)
    return usage


def all_reduce_ema_stats(counts: torch.Tensor, dw: torch.Tensor, group=None) -> None:
    """Sum the per-batch EMA statistics over the ranks, in place: counts [K] int64 and dw [K, D] fp32 (opt-in EMA
    extension, lipvq_vae_amd.ema; 8 KiB + 256 KiB at K = 1024, D = 64)."""
    if counts.dtype != torch.int64 or dw.dtype != torch.float32:
        raise TypeError("EMA statistics must be int64 counts and fp32 sums")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(dw, op=dist.ReduceOp.SUM, group=group)


def all_reduce_gradients(params: Iterable[torch.nn.Parameter], n_local: int, n_global: int, group=None) -> None:
    """Replace every .grad by the gradient of the GLOBAL-mean loss: sum_r (n_r / n_global) * grad_r.
    One flat buffer, one all-reduce."""
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    w = float(n_local) / float(n_global)
    flat = torch.cat([p.grad.reshape(-1) for p in ps]).mul_(w)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


class RcclCounts:
    """The C-ABI binding of the path's collective (include/lipvq.h: lipvq_comm_* / lipvq_allreduce_counts): an RCCL
    communicator owned by the tokenizer library, for callers that do not route the histogram through
    ``torch.distributed``.  Rank 0 draws the 128-byte unique id, ``torch.distributed`` (any backend, used only as the
    out-of-band channel) hands it to the other ranks, and every rank joins with its CURRENT device.

    ``all_reduce(usage)`` enqueues the in-place sum on a side stream ordered after the current stream and returns an
    event; ``wait(event)`` makes the current stream wait for it -- the reduction of batch k overlaps the tokenize
    launch of batch k+1 the same way ``dist.all_reduce(async_op=True)`` does in bench.py."""

    def __init__(self, group=None):
        import ctypes as C

        from . import _capi
        self._capi, self._C = _capi, C
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            raw = (C.c_ubyte * 128)()
            _capi.check(_capi.lib.lipvq_comm_unique_id(C.cast(raw, C.c_void_p)), "lipvq_comm_unique_id")
            ident = torch.tensor(list(raw), dtype=torch.uint8)
        if self.world > 1:
            box = [ident.tolist()]
            dist.broadcast_object_list(box, src=0, group=group)      # the store carries it: works for nccl and gloo groups
            ident = torch.tensor(box[0], dtype=torch.uint8)
        raw = (C.c_ubyte * 128)(*ident.tolist())
        comm = C.c_void_p()
        _capi.check(_capi.lib.lipvq_comm_init(C.byref(comm), C.cast(raw, C.c_void_p), self.rank, self.world), "lipvq_comm_init")
        self._comm = comm
        self._side = torch.cuda.Stream()

    def all_reduce(self, usage: torch.Tensor) -> "torch.cuda.Event":
        if not (usage.is_cuda and usage.dtype == torch.int64 and usage.is_contiguous()):
            raise TypeError("RcclCounts.all_reduce: a contiguous int64 CUDA tensor is required")
        self._side.wait_stream(torch.cuda.current_stream())
        self._capi.check(self._capi.lib.lipvq_allreduce_counts(usage.data_ptr(), usage.numel(), self._comm,
                                                               self._side.cuda_stream), "lipvq_allreduce_counts")
        usage.record_stream(self._side)
        ev = torch.cuda.Event()
        ev.record(self._side)
        return ev

    def all_reduce_f32(self, buf: torch.Tensor) -> "torch.cuda.Event":
        if not (buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()):
            raise TypeError("RcclCounts.all_reduce_f32: a contiguous fp32 CUDA tensor is required")
        self._side.wait_stream(torch.cuda.current_stream())
        self._capi.check(self._capi.lib.lipvq_allreduce_f32(buf.data_ptr(), buf.numel(), self._comm, self._side.cuda_stream),
                         "lipvq_allreduce_f32")
        buf.record_stream(self._side)
        ev = torch.cuda.Event()
        ev.record(self._side)
        return ev

    @staticmethod
    def wait(event) -> None:
        torch.cuda.current_stream().wait_event(event)

    def close(self) -> None:
        if self._comm is not None and self._comm.value:
            torch.cuda.synchronize()
            self._capi.check(self._capi.lib.lipvq_comm_destroy(self._comm), "lipvq_comm_destroy")
        self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedTokenizer:
    """Tokenise this rank's share of a batch and keep the global code-usage statistics.

    ``tokenizer`` is anything with ``tokenize(x[N,A]) -> (indices[N], z_latent[N,D])`` and a
    ``code_usage`` int64 tensor -- normally ``lipvq_vae_amd.tokenizer.LLFQVAE_V4`` on this rank's GPU.
    """

    def __init__(self, tokenizer, group=None):
        self.tokenizer = tokenizer
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def tokenize(self, actions: torch.Tensor, reduce_usage: bool = True):
        """actions [B, T, A] (the full batch) -> (indices [B_local, T], z_latent [B_local, T, D]) of this rank's rows.

        ``tokenizer.code_usage`` is a CUMULATIVE histogram (the kernels add into it on every call), so only THIS
        batch's counts cross the ranks: the delta of the call is summed over the ranks and added to what the buffer
        held before, which is already global.  After any number of calls every rank holds the histogram of all rows
        of all batches (reducing the buffer itself would re-multiply earlier batches by the world size each call)."""
        s, e = shard_bounds(actions.shape[0], self.rank, self.world)
        local = actions[s:e]
        b, t = local.shape[0], local.shape[1]
        usage = self.tokenizer.code_usage
        before = usage.clone() if reduce_usage else None
        idx, z = self.tokenizer.tokenize(local.reshape(b * t, -1))
        if reduce_usage:
            usage = self.tokenizer.code_usage
            delta = usage - before
            all_reduce_usage(delta, self.group)
            usage.copy_(before + delta)
        return idx.reshape(b, t), z.reshape(b, t, -1)
